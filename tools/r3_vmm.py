"""oligo k=4 writing to outputs made of separately created physical granules mapped in shuffled order (tools/ubench/vmm_alloc.hip)
against plain allocations: can the fast placement class be had on purpose?"""
import os, sys, pathlib, ctypes as C
os.environ["KT_KNOBS_LIVE"] = "1"
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
import torch
from kmertools_amd import device
vmm = C.CDLL(str(root / "tools/ubench/libvmm_alloc.so"))
vmm.vmm_alloc.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
vmm.vmm_free.argtypes = [C.c_void_p]
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
def run(label, ptrs):
    for per in ("32", "96"):
        os.environ["KT_OLIGO_OVERSUB"] = per
        print("%-34s %s/slot: " % (label, per) + " ".join("%.3f" % timed(lambda: ctx.oligo(bases, offsets, n, 4, p)) for p in ptrs), flush=True)
plain = [torch.empty((n, 136), dtype=torch.float64, device="cuda") for _ in range(4)]
run("plain (torch / hipMalloc)", [t.data_ptr() for t in plain])
del plain; torch.cuda.empty_cache()
for chunk_mb, shuffle in ((2, 1), (2, 0), (32, 1), (256, 1), (1024, 1)):
    hs, ps = [], []
    for _ in range(3):
        p, h = C.c_void_p(), C.c_void_p()
        rc = vmm.vmm_alloc(size, chunk_mb << 20, shuffle, C.byref(p), C.byref(h))
        if rc: print("vmm_alloc failed", rc); break
        hs.append(h); ps.append(p.value)
    if ps: run("vmm %4d MB chunks%s" % (chunk_mb, " shuffled" if shuffle else ""), ps)
    for h in hs: vmm.vmm_free(h)
