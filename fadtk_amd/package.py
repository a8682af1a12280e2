"""`python -m fadtk_amd.package ...` -- see fadtk_amd/cli.py:package_main."""
from .cli import package_main as main

if __name__ == "__main__":
    main()
