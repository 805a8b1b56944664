"""pykmertools - the reference's Python module by name, running on the MI355X.

`import pykmertools as kt` is what users of anuradhawick/kmertools write (tests/test_oligo.py:1-5 of the reference;
module registered at pip/src/lib.rs:31-40 with OligoComputer, CgrComputer, KmerGenerator, MinimiserGenerator,
run_cli and the utils submodule).  Everything here is re-exported from kmertools_amd.pykmertools, which drives
libkmertools_hip.so through the C ABI; there is no CPU fallback.
"""
from kmertools_amd.pykmertools import (CgrComputer, KmerGenerator, MinimiserGenerator, OligoComputer,  # noqa: F401
                                       run_cli)
from . import utils  # noqa: F401

__all__ = ["OligoComputer", "CgrComputer", "KmerGenerator", "MinimiserGenerator", "run_cli", "utils"]
