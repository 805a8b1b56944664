"""oligo k=4 writing to outputs allocated with hipExtMallocWithFlags(hipDeviceMallocContiguous) against plain hipMalloc:
does physical contiguity decide the placement class?"""
import os, sys, pathlib, ctypes as C
os.environ["KT_KNOBS_LIVE"] = "1"
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
n, L = 10_000_000, 150
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
def timed(fn, reps=10, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps): fn()
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
size = n * 136 * 8
def alloc(flag):
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), size, flag) if flag is not None else hip.hipMalloc(C.byref(p), size)
    return p.value if rc == 0 else None
p = alloc(4)
for R in (26, 32, 40, 46, 53):
    row = []
    for per in (32, 48, 64, 80, 96, 112, 128, 160, 200):
        os.environ["KT_OLIGO_R"] = str(R); os.environ["KT_OLIGO_OVERSUB"] = str(per)
        row.append("%d: %.3f" % (per, timed(lambda: ctx.oligo(bases, offsets, n, 4, p))))
    print("contiguous R=%d  %s" % (R, "  ".join(row)), flush=True)
