"""`python -m fadtk_amd.embeds ...` -- see fadtk_amd/cli.py:embeds_main."""
from .cli import embeds_main as main

if __name__ == "__main__":
    main()
