"""ctypes binding of libkmertools_hip.so (the C ABI declared in include/kmertools_hip.h).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is usable,
the calls raise.  Nothing here imports the parity oracle.
"""
import ctypes as C
import os
import pathlib
import subprocess

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = pathlib.Path(os.environ.get("KT_LIB", _HERE / "libkmertools_hip.so"))  # KT_LIB: A/B builds

KT_OK = 0
KT_ERR_ARG, KT_ERR_HIP, KT_ERR_NOMEM, KT_ERR_FULL, KT_ERR_NODEVICE, KT_ERR_BADNT = 1, 2, 3, 4, 5, 6
KT_MEM_HOST, KT_MEM_DEVICE = 0, 1
KT_F64, KT_F32, KT_U32 = 0, 1, 2
KT_EMPTY_KEY = 0xFFFFFFFFFFFFFFFF

# every symbol include/kmertools_hip.h declares: name -> (restype, argtypes)
_vp, _u64, _u32, _i = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
SYMBOLS = {
    "kt_version": (_i, []),
    "kt_last_error": (C.c_char_p, []),
    "kt_device_count": (_i, [C.POINTER(_i)]),
    "kt_ctx_create": (_i, [_i, _vp, _i, C.POINTER(_vp)]),
    "kt_ctx_destroy": (_i, [_vp]),
    "kt_device_memory": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "kt_device_alloc_placed": (_i, [_vp, _u64, _i, _i, _vp, _vp, C.POINTER(_vp), C.POINTER(C.c_double), C.POINTER(_i),
                               C.POINTER(_i)]),
    "kt_device_free": (_i, [_vp, _vp]),
    "kt_host_register": (_i, [_vp, _vp, C.c_size_t]),
    "kt_host_unregister": (_i, [_vp, _vp]),
    "kt_ctx_sync": (_i, [_vp]),
    "kt_bins": (_i, [_i, _i, C.POINTER(_u64)]),
    "kt_pos_map": (_i, [_i, _vp, _vp, C.POINTER(_u32)]),
    "kt_rev_comp": (_u64, [_u64, _i]),
    "kt_numeric_to_kmer": (_i, [_u64, _i, C.c_char_p]),
    "kt_kmer_to_numeric": (_i, [C.c_char_p, _u64, C.POINTER(_u64), C.POINTER(_u64)]),
    "kt_cgr_coords": (_i, [_i, C.c_double, _vp]),
    "kt_kmers": (_i, [_vp, _vp, _vp, _u64, _i, _vp, _vp, _vp, _i]),
    "kt_oligo_batch": (_i, [_vp, _vp, _vp, _u64, _i, _i, _i, _i, _i, _vp, _i]),
    "kt_oligo_tuning": (_i, [_vp, _i]),
    "kt_oligo_launch_info": (_i, [_vp, C.POINTER(_u32), C.POINTER(_i), C.POINTER(C.c_double)]),
    "kt_selftest_quotient": (_i, [_vp, _u32, _u32, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "kt_ctr_create": (_i, [_vp, _i, _u64, C.POINTER(_vp)]),
    "kt_ctr_destroy": (_i, [_vp]),
    "kt_ctr_clear": (_i, [_vp]),
    "kt_ctr_add_reads": (_i, [_vp, _vp, _vp, _u64, _i]),
    "kt_ctr_add_reads_part": (_i, [_vp, _vp, _vp, _u64, _i, _u32, _u32]),
    "kt_ctr_add_pairs": (_i, [_vp, _vp, _vp, _u64, _i]),
    "kt_ctr_size": (_i, [_vp, C.POINTER(_u64)]),
    "kt_ctr_capacity": (_i, [_vp, C.POINTER(_u64)]),
    "kt_ctr_export": (_i, [_vp, _vp, _vp, _u64, C.POINTER(_u64), _i]),
    "kt_ctr_export_target": (_i, [_vp, _vp, _vp, _u64]),
    "kt_ctr_export_stage": (_i, [_vp, C.POINTER(_u64)]),
    "kt_ctr_export_fetch": (_i, [_vp, _u64, _u64, _vp, _vp]),
    "kt_cgr_points": (_i, [_vp, _vp, _vp, _u64, C.c_double, _vp, _vp, _i]),
    "kt_minimisers": (_i, [_vp, _vp, _vp, _u64, _u64, _i, _vp, _vp, _vp, _vp, _u64, C.POINTER(_u64), _i]),
    "kt_cov_batch": (_i, [_vp, _vp, _vp, _u64, _u64, _u64, _i, _i, _vp, _i]),
    "kt_cov_batch_part": (_i, [_vp, _vp, _vp, _u64, _u64, _u64, _vp, _i, _u32, _u32]),
    "kt_ctr_lookup": (_i, [_vp, _vp, _u64, _vp, _i]),
    "kt_ctr_route": (_i, [_vp, _vp, _vp, _u64, _i, _i, _vp, _vp, _i]),
    "kt_owner_of": (_u32, [_u64, _u32]),
    "kt_rccl_unique_id": (_i, [_vp]),
    "kt_sharded_create_rccl": (_i, [_vp, _i, _u64, _u64, _i, _i, _vp, C.POINTER(_vp)]),
    "kt_sharded_create_host": (_i, [_vp, _i, _u64, _u64, _i, _i, _vp, _vp, C.POINTER(_vp)]),
    "kt_sharded_destroy": (_i, [_vp]),
    "kt_sharded_clear": (_i, [_vp]),
    "kt_sharded_add_reads": (_i, [_vp, _vp, _vp, _u64, _i]),
    "kt_sharded_finalize": (_i, [_vp]),
    "kt_sharded_table": (_i, [_vp, C.POINTER(_vp)]),
    "kt_sharded_exchanged_bytes": (_i, [_vp, C.POINTER(_u64)]),
    "kt_sharded_comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "kt_sharded_create_local": (_i, [_vp, _i, _u64, _u64, _i, _i, C.POINTER(_vp)]),
    "kt_sharded_connect_rccl": (_i, [_vp, _vp]),
    "kt_sharded_connect_host": (_i, [_vp, _vp, _vp]),
    "kt_sharded_owner_of": (_i, [_vp, _u64, C.POINTER(_u32)]),
    "kt_sharded_route_stats": (_i, [_vp, C.POINTER(_u32), _vp, _vp]),
    "kt_shard_minimiser": (_i, [_i, C.POINTER(_u32), C.POINTER(_u32)]),
    "kt_shard_owner_of": (_u32, [_u64, _i, _u32]),
    "kt_synth_reads": (_i, [_vp, _u64, _u64, _u64, _u32, _i, _u64, _vp, _vp]),
}


ALLTOALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)  # kt_alltoall_fn
PROBE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)                              # kt_probe_fn


class KmertoolsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libkmertools_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def build(force=False):
    """Compile libkmertools_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    srcdir = _HERE / "csrc"
    if force:
        subprocess.check_call(["make", "-C", str(srcdir), "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", str(srcdir), "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                "%s is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C kmertools_amd/csrc`; there is no CPU fallback" % LIB_PATH)
        # torch bundles its own libamdhip64; importing it first makes this library resolve to
        # the same HIP runtime instance, so torch's streams/pointers are valid here (two
        # runtimes in one process cannot both see the GPU).  Without torch, /opt/rocm's is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    s = lib().kt_last_error()
    return s.decode("utf-8", "replace") if s else ""


def check(rc):
    if rc != KT_OK:
        raise KmertoolsError(rc, last_error())
    return rc
