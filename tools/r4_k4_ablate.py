import os, sys, pathlib
os.environ["KT_KNOBS_LIVE"] = "1"
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
n, L, k = 10_000_000, 150, 4
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
ctx.oligo_tuning(96)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
out = torch.empty((n, 136), dtype=torch.float64, device="cuda")
for dbg in (0, 4, 1, 2, 0, 4):
    os.environ["KT_OLIGO_DEBUG"] = str(dbg)
    for _ in range(3): ctx.oligo(bases, offsets, n, k, out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(10): ctx.oligo(bases, offsets, n, k, out)
    b.record(s); torch.cuda.synchronize()
    print("k=4 debug=%d: %.3f ms per 10 M reads" % (dbg, a.elapsed_time(b) / 10), flush=True)
