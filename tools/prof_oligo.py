#!/usr/bin/env python3
"""Small driver for rocprofv3: a few launches of one hot-path kernel on BASELINE-size input.
usage: prof_oligo.py [oligo|cgr7|ctr31|ctr15] [launches]"""
import os
import sys
import pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device

what = sys.argv[1] if len(sys.argv) > 1 else "oligo"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
L = 150
if what in ("oligo", "cgr7"):
    k, dtype, n = (4, "f64", 10_000_000) if what == "oligo" else (7, "f32", 1_000_000)
    n = int(os.environ.get("N", n))
    bases = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_reads(1, n, L, bases, offsets)
    out = torch.empty((n, device.bins(k)), dtype=torch.float64 if dtype == "f64" else torch.float32, device="cuda")
    for _ in range(reps):
        ctx.oligo(bases, offsets, n, k, out, dtype=dtype)
else:
    k = 31 if what == "ctr31" else 15
    n = int(os.environ.get("N", 5_000_000))
    bases = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_reads(1, n, L, bases, offsets)
    ctr = device.Counter(ctx, k, 1 << (2 * n * (L - k + 1) - 1).bit_length())
    for _ in range(reps):
        ctr.clear()
        ctr.add_reads(bases, offsets, n)
torch.cuda.synchronize()
print("done", what)
