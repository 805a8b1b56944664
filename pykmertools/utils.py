"""pykmertools.utils - `to_acgt` / `to_numeric` (pybindings/src/kmer.rs:49-65, registered as a submodule at :67-76)"""
from kmertools_amd.pykmertools import utils as _u

to_acgt = _u.to_acgt
to_numeric = _u.to_numeric

__all__ = ["to_acgt", "to_numeric"]
