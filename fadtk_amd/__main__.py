"""`python -m fadtk_amd ...` -- see fadtk_amd/cli.py:score_main."""
from .cli import score_main as main

if __name__ == "__main__":
    main()
