"""comp cgr k=7 (8192-bin canonical rows, f32): workgroups per resident slot"""
import os, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
n, L = 1_000_000, 150
k = int(sys.argv[1]) if len(sys.argv) > 1 else 7
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
out = torch.empty((n, device.bins(k, True)), dtype=torch.float32, device="cuda")
for ov in (32, 2, 4, 8, 16, 32):
    os.environ["KT_OLIGO_OVERSUB"] = str(ov)
    fn = lambda: ctx.oligo(bases, offsets, n, k, out, dtype="f32")
    for _ in range(6): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(6): fn()
    b.record(s); torch.cuda.synchronize()
    print("k=%d f32 1 M reads, oversub %3d: %.3f ms" % (k, ov, a.elapsed_time(b) / 6), flush=True)
