"""What makes a process 'fast' or 'slow' for oligo k=4 with 32 workgroups per resident slot?  One process, everything held
fixed except one thing at a time: the input copy (bases / offsets allocations), the stream (hardware queue), the
context (LUT allocation), the output allocation."""
import os, sys, pathlib
os.environ["KT_OLIGO_OVERSUB"] = "32"
os.environ["KT_KNOBS_LIVE"] = "1"
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
n, L = 10_000_000, 150
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
out = torch.empty((n, 136), dtype=torch.float64, device="cuda")

def timed(fn, stream=s, reps=20, warm=22):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps): fn()
    b.record(stream); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

def both(label, fn, stream=s):
    r = []
    for setting in ("32", "96"):
        os.environ["KT_OLIGO_OVERSUB"] = setting
        r.append(timed(fn, stream))
    print("%-60s 32: %.3f ms   96: %.3f ms" % (label, r[0], r[1]), flush=True)

both("baseline", lambda: ctx.oligo(bases, offsets, n, 4, out))
keep = []
for i in range(5):
    b2 = bases.clone(); o2 = offsets.clone(); keep.append((b2, o2))
    both("input copy %d (bases @%x)" % (i, b2.data_ptr()), lambda: ctx.oligo(b2, o2, n, 4, out))
for i in range(4):
    o3 = torch.empty((n, 136), dtype=torch.float64, device="cuda"); keep.append(o3)
    both("output allocation %d (@%x)" % (i, o3.data_ptr()), lambda: ctx.oligo(bases, offsets, n, 4, o3))
for i in range(4):
    st = torch.cuda.Stream()
    c2 = device.Context(0, stream=st.cuda_stream); keep.append((st, c2))
    with torch.cuda.stream(st):
        both("new stream + context %d" % i, lambda: c2.oligo(bases, offsets, n, 4, out), st)
both("baseline again", lambda: ctx.oligo(bases, offsets, n, 4, out))
