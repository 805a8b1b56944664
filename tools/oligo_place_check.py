"""oligo k=4 (10 M reads -> 10.9 GB of rows) against where the output lies: slices of one 96 GiB allocation at different
offsets, and outputs of smaller spans written by several launches (see tools/oligo_ring_check.py)."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
n, L = 10_000_000, 150
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)

def timed(fn, reps=20):
    for _ in range(22): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps): fn()
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

big = torch.empty(96 << 30, dtype=torch.uint8, device="cuda")
rows = n * 136 * 8
for off_gib in (0, 1, 7, 16, 33, 50, 64, 80):
    out = big[off_gib << 30:(off_gib << 30) + rows].view(torch.float64).view(n, 136)
    print("output at +%2d GiB of a 96 GiB block: %.3f ms" % (off_gib, timed(lambda: ctx.oligo(bases, offsets, n, 4, out))), flush=True)
for shift in (256, 4096, 65536, 1 << 20):
    out = big[shift:shift + rows].view(torch.float64).view(n, 136)
    print("output at +%d bytes: %.3f ms" % (shift, timed(lambda: ctx.oligo(bases, offsets, n, 4, out))), flush=True)
for parts in (2, 5, 10, 20):
    m = n // parts
    offs = [(offsets[i * m:(i + 1) * m + 1] - offsets[i * m]).contiguous() for i in range(parts)]
    ring = big[:m * 136 * 8].view(torch.float64).view(m, 136)
    def many():
        for i in range(parts):
            ctx.oligo(bases[i * m * L:(i + 1) * m * L], offs[i], m, 4, ring)
    print("%2d launches into one %.2f GB output: %.3f ms per 10 M reads" % (parts, m * 1088 / 1e9, timed(many)), flush=True)
