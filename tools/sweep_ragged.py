#!/usr/bin/env python3
"""oligo kernel on variable-length reads (general path): Gbases/s vs the equal-length fast path"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
s = torch.cuda.current_stream(); ctx = device.Context(0, stream=s.cuda_stream)
n, k = 10_000_000, 4
for name, lo, hi in [("uniform 150", 150, 151), ("100-200", 100, 201), ("50-250", 50, 251), ("20-5000 long tail", 20, 5001)]:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    nn = n if hi < 1000 else 600_000
    lens = torch.randint(lo, hi, (nn,), device="cuda", generator=g, dtype=torch.int64)
    offsets = torch.zeros(nn + 1, dtype=torch.int64, device="cuda"); offsets[1:] = torch.cumsum(lens, 0)
    total = int(offsets[-1])
    bases = torch.empty(total, dtype=torch.uint8, device="cuda")
    ctx.synth_reads(3, total // 150 + 1, 150, torch.empty((total // 150 + 1) * 150, dtype=torch.uint8, device="cuda"))
    tmp = torch.empty((total // 150 + 1) * 150, dtype=torch.uint8, device="cuda"); ctx.synth_reads(3, total // 150 + 1, 150, tmp); bases = tmp[:total].contiguous()
    out = torch.empty((nn, 136), dtype=torch.float64, device="cuda")
    for _ in range(2): ctx.oligo(bases, offsets, nn, k, out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(5): ctx.oligo(bases, offsets, nn, k, out)
    b.record(s); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print("%-22s reads=%d bases=%.2fG  %.3f ms  %.1f Gbases/s  %.0f GB/s" % (name, nn, total / 1e9, ms, total / ms / 1e6, (total + nn * 1088) / ms / 1e6), flush=True)
