"""comp cgr k=7 / k=6 (f32 rows, 1 M reads per launch over a 10 M-read input, so that the input is not cache resident):
the big-row kernel with and without its producer wave (KT_OLIGO_PW = smallest k that gets one; 8 = none), and - with a
KT_OLIGO_ABLATION=1 variant through KT_LIB - phases switched off (KT_OLIGO_DEBUG bits: 1 no counting, 2 no row stores,
4 constants instead of the chunk loads).  usage: r4_k7_ablate.py [debug values ...]"""
import os, sys, pathlib
os.environ["KT_KNOBS_LIVE"] = "1"
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
n, L, B = 10_000_000, 150, 1_000_000
dbgs = [int(x) for x in sys.argv[1:]] or [0]
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
offs = [(offsets[i * B:(i + 1) * B + 1] - offsets[i * B]).contiguous() for i in range(n // B)]
for k in (7, 6):
    out = torch.empty((B, device.bins(k, True)), dtype=torch.float32, device="cuda")
    def step():
        for i in range(n // B):
            ctx.oligo(bases[i * B * L:], offs[i], B, k, out, dtype="f32")
    for rep in range(2):
        for pw in (8, 6):
            for dbg in dbgs:
                os.environ["KT_OLIGO_PW"] = str(pw); os.environ["KT_OLIGO_DEBUG"] = str(dbg)
                step(); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(s); step(); step(); b.record(s); torch.cuda.synchronize()
                print("k=%d producer wave %s debug=%d: %.3f ms per 1 M reads" % (k, "yes" if pw <= k else "no ", dbg, a.elapsed_time(b) / 20), flush=True)
    del out
