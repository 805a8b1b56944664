import os, sys, pathlib
sys.path.insert(0, ".")
import torch
from kmertools_amd import device
n, L = 10_000_000, 150
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
for k in (3, 5, 4):
    out = torch.empty((n, device.bins(k, True)), dtype=torch.float64, device="cuda")
    for ov in (32, 96, 32, 96):
        os.environ["KT_OLIGO_OVERSUB"] = str(ov)
        fn = lambda: ctx.oligo(bases, offsets, n, k, out)
        for _ in range(22): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(20): fn()
        b.record(s); torch.cuda.synchronize()
        print("k=%d oversub %3d: %.3f ms" % (k, ov, a.elapsed_time(b) / 20), flush=True)
    del out
