"""GPU box: the ctr step of bench.py (clear + add_reads + size + export into the export target) with no output check -
for ablation builds (KT_LIB=variants/lib*.so, KT_BUILD_DBG bits) whose results are wrong on purpose.  Prints ms per step;
run under rocprofv3 --kernel-trace --stats (tools/r4_abl.sh) for per-kernel times."""
import argparse, os, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device, dist as ktdist

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--reads", type=int, default=25_000_000)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--cap-factor", type=float, default=1.9)
a = ap.parse_args()
L = 150
ctx = device.Context(0, stream=torch.cuda.current_stream().cuda_stream)
n = a.reads
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda")
offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(0x6b6d6572, n, L, bases, offsets, noise=False)
kpr = L - a.k + 1
max_distinct = min(n * kpr, (4 ** a.k + 2 ** a.k) // 2)
cap = max(1 << 20, int(a.cap_factor * max_distinct))
counter = ktdist.ShardedCounter(ctx, a.k, cap, group=None, max_batch_bases=n * L)
xk = torch.empty(max_distinct, dtype=torch.int64, device="cuda")
xc = torch.empty(max_distinct, dtype=torch.int32, device="cuda")
counter.table.export_target(xk, xc, max_distinct)
ms = []
for i in range(a.steps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    counter.clear()
    counter.add_reads(bases, offsets, n)
    counter.finalize()
    try:
        d = counter.size_local()
        counter.table.export(xk, xc, max_distinct)
    except Exception as e:  # ablation builds may leave the table in a state the export refuses
        d = -1
    torch.cuda.synchronize()
    ms.append((time.perf_counter() - t0) * 1e3)
try:  # a timing build (KT_ABLATION): cycles of thread 0 per phase, summed over workgroups and steps
    import ctypes
    from kmertools_amd import _lib
    f = _lib.lib().kt_dbg_phases
    buf = (ctypes.c_ulonglong * 16)()
    if f(buf) == 0 and sum(buf):
        for name, lo in (("build", 0), ("scatter1y", 8)):
            tot = float(sum(buf[lo:lo + 8]))
            if tot:
                print("%s phases (share of thread 0's cycles): " % name + " ".join("%d:%.1f%%" % (i, 100 * buf[lo + i] / tot) for i in range(8)))
except AttributeError:
    pass
print("k=%d reads=%d distinct=%d ms/step %s" % (a.k, n, d, " ".join("%.2f" % x for x in ms[1:])))
