"""Does the per-array measurement of oligo k=4 pick what the steady state says?  Several output arrays in one process:
steady-state ms with 32 / 96 / 200 workgroups per resident slot fixed, then what a fresh context measures and picks
for the same array, and the ms it then runs at."""
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
def timed(fn, reps=20, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps): fn()
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
outs = [torch.empty((n, 136), dtype=torch.float64, device="cuda") for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10)]
for i, o in enumerate(outs):
    fixed = {}
    for per in (32, 96, 200):
        os.environ["KT_OLIGO_OVERSUB"] = str(per)
        fixed[per] = timed(lambda: ctx.oligo(bases, offsets, n, 4, o))
    os.environ.pop("KT_OLIGO_OVERSUB")
    c = device.Context(0, stream=s.cuda_stream)
    for _ in range(10):                      # launches in bursts with a sync between, as a pipeline would
        for _ in range(6): c.oligo(bases, offsets, n, 4, o)
        torch.cuda.synchronize()
    info = c.oligo_launch_info()
    t = timed(lambda: c.oligo(bases, offsets, n, 4, o))
    best = min(fixed, key=fixed.get)
    print("array %2d: fixed %s  -> measured %s picked %d, runs at %.3f ms %s" % (
        i, " ".join("%d: %.3f" % kv for kv in fixed.items()),
        " ".join("%d: %.3f" % (k, v * n / 1e6) for k, v in info["ns_per_read"].items()), info["wgs_per_slot"], t,
        "" if fixed[info["wgs_per_slot"]] <= 1.012 * fixed[best] else "<-- not the best"), flush=True)
    c.close()
