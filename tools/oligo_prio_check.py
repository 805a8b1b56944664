"""oligo k=4: store-phase wave priority (KT_OLIGO_DEBUG bit 16 = off) x workgroups per resident slot"""
import os, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
n, L = 10_000_000, 150
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
out = torch.empty((n, 136), dtype=torch.float64, device="cuda")
def timed(reps=20):
    fn = lambda: ctx.oligo(bases, offsets, n, 4, out)
    for _ in range(22): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps): fn()
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for ov in (32, 96):
    for dbg in ("", "16"):
        os.environ["KT_OLIGO_OVERSUB"] = str(ov)
        if dbg: os.environ["KT_OLIGO_DEBUG"] = dbg
        else: os.environ.pop("KT_OLIGO_DEBUG", None)
        print("oversub %3d debug %2s: %.3f ms" % (ov, dbg or "0", timed()), flush=True)
