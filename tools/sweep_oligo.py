#!/usr/bin/env python3
"""Times the oligo kernel under different launch tunables (KT_OLIGO_R / _STAGE / _OVERSUB)."""
import itertools
import os
import sys
import pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device

k = int(os.environ.get("K", 4)); dtype = os.environ.get("DT", "f64")
n, L = int(os.environ.get("N", 10_000_000)), 150
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda")
offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
bins = device.bins(k)
out = torch.empty((n, bins), dtype=torch.float64 if dtype == "f64" else torch.float32, device="cuda")
esz = 8 if dtype == "f64" else 4
Rs = [int(x) for x in os.environ.get("RS", "8,16,32,64").split(",")]
OV = [int(x) for x in os.environ.get("OV", "1,2,4").split(",")]
for R, ov in itertools.product(Rs, OV):
    os.environ["KT_OLIGO_R"] = str(R); os.environ["KT_OLIGO_OVERSUB"] = str(ov)
    for _ in range(2): ctx.oligo(bases, offsets, n, k, out, dtype=dtype)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(5): ctx.oligo(bases, offsets, n, k, out, dtype=dtype)
    b.record(s); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print("R=%3d oversub=%d  %.3f ms  %.1f Gbases/s  %.0f GB/s" % (R, ov, ms, n * L / ms / 1e6, n * (L + bins * esz) / ms / 1e6), flush=True)
# memset reference for the store roofline
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out.zero_(); torch.cuda.synchronize(); a.record(s)
for _ in range(5): out.zero_()
b.record(s); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
print("torch zero_ of the output: %.3f ms  %.0f GB/s" % (ms, out.numel() * esz / ms / 1e6))
