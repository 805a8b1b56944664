"""oligo k=4 (static shares, 32 and 96 workgroups per resident slot) writing to each of ~22 separately allocated 10.9 GB
outputs that together cover most of the 288 GB: does the kernel's 'kind' go with where in the memory the output lies?"""
import os, sys, pathlib
os.environ["KT_KNOBS_LIVE"] = "1"
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
n, L = 10_000_000, 150
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
def timed(fn, reps=10, warm=6):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps): fn()
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
outs = []
for i in range(23):
    try:
        outs.append(torch.empty((n, 136), dtype=torch.float64, device="cuda"))
    except Exception as e:
        print("allocation %d failed: %s" % (i, str(e)[:80])); break
free, total = torch.cuda.mem_get_info()
print("%d outputs allocated, %.1f of %.1f GB free" % (len(outs), free / 1e9, total / 1e9), flush=True)
def rate(o):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    o.zero_(); a.record(s)
    for _ in range(5): o.zero_()
    b.record(s); torch.cuda.synchronize()
    return 5 * o.numel() * 8 / (a.elapsed_time(b) * 1e-3) / 1e12
print("zero_ TB/s:      " + " ".join("%.3f" % rate(o) for o in outs), flush=True)
print("address >> 30:   " + " ".join("%5d" % (o.data_ptr() >> 30 & 0xffff) for o in outs), flush=True)
pats = [("static 32", 32, 0, 0), ("static 200", 200, 0, 0), ("static 32 late loads", 32, 0, 3), ("static 96", 96, 0, 0), ("static 96 late loads", 96, 0, 3)]
for label, per, tick, exp in pats:
    os.environ["KT_OLIGO_OVERSUB"] = str(per); os.environ["KT_OLIGO_TICKETS"] = str(tick); os.environ["KT_OLIGO_EXP"] = str(exp)
    print("%-22s " % label + " ".join("%.3f" % timed(lambda: ctx.oligo(bases, offsets, n, 4, o)) for o in outs), flush=True)
