"""Is the slow kind of box slow because of WHERE the 10.9 GB of output lands (pages / TLB reach)?  The same 10 M reads
written (a) to one 10.9 GB array, (b) in ten launches of 1 M reads into the same 1.09 GB array, (c) like (a) but into an
array allocated after 150 GB of other allocations were made and freed."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
n, L = 10_000_000, 150
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
m = n // 10
off_small = [(offsets[i * m:(i + 1) * m + 1] - offsets[i * m]).contiguous() for i in range(10)]

def timed(fn, reps=30):
    for _ in range(25): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps): fn()
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

out = torch.empty((n, 136), dtype=torch.float64, device="cuda")
print("(a) one launch, 10.9 GB output: %.3f ms per 10 M reads" % timed(lambda: ctx.oligo(bases, offsets, n, 4, out)))
ring = out[:m]
def ten():
    for i in range(10):
        ctx.oligo(bases[i * m * L:(i + 1) * m * L], off_small[i], m, 4, ring)
print("(b) ten launches into one 1.09 GB output: %.3f ms per 10 M reads" % timed(ten))
print("(a) again: %.3f ms" % timed(lambda: ctx.oligo(bases, offsets, n, 4, out)))
del out, ring
torch.cuda.empty_cache()
junk = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(150)]
del junk[::2]
out2 = torch.empty((n, 136), dtype=torch.float64, device="cuda")
print("(c) output allocated between other live 1 GiB blocks: %.3f ms" % timed(lambda: ctx.oligo(bases, offsets, n, 4, out2)))
