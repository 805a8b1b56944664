"""`python -m pykmertools ...` = the kmertools command line (pip/src/lib.rs:11-18)"""
import sys

from . import run_cli

sys.exit(run_cli())
