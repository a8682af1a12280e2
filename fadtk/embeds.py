from fadtk_amd.embeds import main

if __name__ == "__main__":
    main()
