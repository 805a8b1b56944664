"""oligo k=4, one process: which input's placement moves the kernel - bases or offsets - and do shifts inside one
allocation do it too?  (tools/r3_kind_test.py found input copies 4 % apart.)"""
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

def timed(fn, reps=20, warm=12):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps): fn()
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

def show(label, b, o):
    print("%-70s %.3f ms" % (label, timed(lambda: ctx.oligo(b, o, n, 4, out))), flush=True)

show("baseline", bases, offsets)
bs = [bases.clone() for _ in range(6)]
os_ = [offsets.clone() for _ in range(6)]
for i in range(6):
    show("bases copy %d (@%x), offsets original" % (i, bs[i].data_ptr()), bs[i], offsets)
for i in range(6):
    show("bases original, offsets copy %d (@%x)" % (i, os_[i].data_ptr()), bases, os_[i])
big = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
for sh in (0, 4096, 1 << 16, 1 << 21, 1 << 24, 1 << 27, 1 << 30, 3 << 30, 6 << 30):
    v = big[sh:sh + n * L]; v.copy_(bases)
    show("bases at +%d of an 8 GiB block (@%x)" % (sh, v.data_ptr()), v, offsets)
show("baseline again", bases, offsets)
