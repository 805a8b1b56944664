"""comp cgr k=7 (8192-bin canonical rows, f32, 1 M reads per launch over a 10 M-read input so that the input is not
cache-resident): reads per tile R x workgroups per resident slot"""
import os, sys, pathlib
os.environ["KT_KNOBS_LIVE"] = "1"
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
n, L, B = 10_000_000, 150, 1_000_000
k = int(sys.argv[1]) if len(sys.argv) > 1 else 7
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
out = torch.empty((B, device.bins(k, True)), dtype=torch.float32, device="cuda")
offs = [(offsets[i * B:(i + 1) * B + 1] - offsets[i * B]).contiguous() for i in range(n // B)]
def step():
    for i in range(n // B):
        ctx.oligo(bases[i * B * L:], offs[i], B, k, out, dtype="f32")
for R in (4, 3, 2, 1):
    row = []
    for ov in (2, 4, 8, 16, 32):
        os.environ["KT_OLIGO_R"] = str(R); os.environ["KT_OLIGO_OVERSUB"] = str(ov)
        step(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); step(); step(); b.record(s); torch.cuda.synchronize()
        row.append("%d/slot %.3f" % (ov, a.elapsed_time(b) / 20))
    print("k=%d R=%d: %s ms per 1 M reads" % (k, R, "  ".join(row)), flush=True)
