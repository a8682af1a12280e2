from fadtk_amd.__main__ import main

if __name__ == "__main__":
    main()
