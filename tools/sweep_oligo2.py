#!/usr/bin/env python3
"""oligo k=4 f64 under tile sizes / wave layouts / oversubscription, 20 warm-up + 20 timed launches each (the kernel
reaches its steady rate after ~20 launches), baseline first and last so that the box's kind shows."""
import itertools, os, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device

n, L, k = 10_000_000, 150, 4
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda")
offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
out = torch.empty((n, 136), dtype=torch.float64, device="cuda")

def run(env):
    for key in ("KT_OLIGO_R", "KT_OLIGO_OVERSUB", "KT_OLIGO_SHAPE"):
        os.environ.pop(key, None)
    os.environ.update(env)
    def once():
        ctx.oligo(bases, offsets, n, k, out)
    for _ in range(20): once()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(20): once()
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / 20

print("baseline %.3f ms" % run({}), flush=True)
for shape, R, ov in itertools.product(("104", "108"), ("24", "32", "40", "48", "56"), ("8", "32")):
    try:
        ms = run({"KT_OLIGO_SHAPE": shape, "KT_OLIGO_R": R, "KT_OLIGO_OVERSUB": ov})
        print("shape %s R %s oversub %s  %.3f ms" % (shape, R, ov, ms), flush=True)
    except Exception as e:
        print("shape %s R %s oversub %s  failed: %s" % (shape, R, ov, str(e)[:60]), flush=True)
print("baseline %.3f ms" % run({}), flush=True)
