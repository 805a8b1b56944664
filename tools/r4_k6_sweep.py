import os, sys
os.environ["KT_KNOBS_LIVE"] = "1"
import pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
n, L, B = 10_000_000, 150, 1_000_000
k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
offs = [(offsets[i * B:(i + 1) * B + 1] - offsets[i * B]).contiguous() for i in range(n // B)]
out = torch.empty((B, device.bins(k, True)), dtype=torch.float32, device="cuda")
def step():
    for i in range(n // B):
        ctx.oligo(bases[i * B * L:], offs[i], B, k, out, dtype="f32")
cases = ([(8, 0, 0)] + [(8, R, ov) for R in (3, 4, 5, 6, 7, 8, 9) for ov in (8, 32)] + [(8, 0, 0)]) if k == 6 else \
        ([(7, 0, 0)] + [(7, R, ov) for R in (4, 3, 2) for ov in (2, 8, 32)] + [(7, 0, 0)])
for pw, R, ov in cases:
    os.environ["KT_OLIGO_PW"] = str(pw); os.environ["KT_OLIGO_R"] = str(R); os.environ["KT_OLIGO_OVERSUB"] = str(ov)
    step(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s); step(); step(); b.record(s); torch.cuda.synchronize()
    print("k=%d pw=%d R=%d oversub=%d: %.3f ms per 1 M reads" % (k, pw, R, ov, a.elapsed_time(b) / 20), flush=True)
