"""`python -m pyradiomics_amd image|batch.csv [mask] [options]` -- see pyradiomics_amd/scripts.py"""
import sys

from .scripts import main

if __name__ == "__main__":
    sys.exit(main())
