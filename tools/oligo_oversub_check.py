"""oligo k=4, 10 M reads: workgroups per resident slot (KT_OLIGO_OVERSUB; 163 = one tile per workgroup at R = 40)"""
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
for ov in (32, 4, 16, 48, 64, 96, 128, 200, 32):
    os.environ["KT_OLIGO_OVERSUB"] = str(ov)
    print("oversub %3d: %.3f ms" % (ov, timed()), flush=True)
