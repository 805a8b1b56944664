"""kmertools_amd - MI355X (gfx950) drop-in for kmertools' k-mer hot path.

  kmertools_amd._lib         ctypes binding of libkmertools_hip.so (include/kmertools_hip.h)
  kmertools_amd.device       Context / Counter objects over the C ABI (host arrays or device tensors)
  kmertools_amd.pykmertools  mirror of the reference's `pykmertools` Python module
  kmertools_amd.dist         hash-prefix sharded counting over torch.distributed (RCCL / gloo)

There is no CPU fallback: every compute call goes through the HIP library and raises
if it (or a GPU) is missing.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
