"""oligo k=4 over six copies of the bases, 12 launches each, in that order - to be run under rocprofv3 --pmc with
--kernel-trace (tools/r3_kind_pmc.sh groups the dispatches in runs of 12)"""
import os, sys, pathlib
os.environ["KT_OLIGO_OVERSUB"] = "96"
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
n, L = 10_000_000, 150
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
out = torch.empty((n, 136), dtype=torch.float64, device="cuda")
bs = [bases] + [bases.clone() for _ in range(7)]
torch.cuda.synchronize()
for b in bs:
    for _ in range(12):
        ctx.oligo(b, offsets, n, 4, out)
    torch.cuda.synchronize()
