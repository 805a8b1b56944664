"""oligo k=4 launch tunables on an output array of the FAST placement class (round 2 swept them on the slow kind only):
picks the fastest of 10 allocations with 32 workgroups per slot, then sweeps reads per tile R x workgroups per slot there,
and the 8-wave workgroup shape."""
import os, sys, pathlib
os.environ["KT_KNOBS_LIVE"] = "1"
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
n, L = 10_000_000, 150
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
def timed(fn, reps=12, warm=6):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps): fn()
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
os.environ["KT_OLIGO_OVERSUB"] = "32"
outs = [torch.empty((n, 136), dtype=torch.float64, device="cuda") for _ in range(10)]
ms = [timed(lambda: ctx.oligo(bases, offsets, n, 4, o)) for o in outs]
out = outs[ms.index(min(ms))]
print("32 per slot on ten arrays: " + " ".join("%.3f" % x for x in ms), flush=True)
del outs
for shape in ("104",):
    os.environ["KT_OLIGO_SHAPE"] = shape
    for R in (26, 27, 33, 34, 38, 39, 40, 41, 42, 46, 47, 53, 54, 60):
        row = []
        for per in (24, 32, 48, 96):
            os.environ["KT_OLIGO_R"] = str(R); os.environ["KT_OLIGO_OVERSUB"] = str(per)
            try:
                row.append("%d/slot %.3f" % (per, timed(lambda: ctx.oligo(bases, offsets, n, 4, out))))
            except Exception as e:
                row.append("%d/slot err" % per)
        print("shape %s R=%2d: %s" % (shape, R, "  ".join(row)), flush=True)
