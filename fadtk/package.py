from fadtk_amd.package import main

if __name__ == "__main__":
    main()
