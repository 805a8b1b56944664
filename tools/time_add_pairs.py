"""times kt_ctr_add_pairs(keys only) into an empty table: bulk path vs atomic path (KT_BULK=0), on the GPU box"""
import os, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from kmertools_amd import device
n, L, k = int(os.environ.get("N", 10_000_000)), 150, 31
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda")
offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(5, n, L, bases, offsets)
keys = torch.empty(n * L, dtype=torch.int64, device="cuda")
oc = torch.empty(64, dtype=torch.int64, device="cuda")
ctx.route(bases, offsets, n, k, 1, keys, oc)
nk = int(oc[0].item())
print("keys", nk)
cap = 1 << (2 * nk - 1).bit_length()
ctr = device.Counter(ctx, k, cap)
for mode in ("1", "0", "1", "0"):
    os.environ["KT_BULK"] = mode
    ts = []
    for rep in range(3):
        ctr.clear()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); ctr.add_pairs(keys, None, nk); b.record(s); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print("KT_BULK=%s add_pairs(keys): %s ms -> %.1f G keys/s, distinct %d" % (mode, ["%.1f" % t for t in ts], nk / min(ts) / 1e6, ctr.size()))
for G in (1, 2, 8):
    ts = []
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); ctx.route(bases, offsets, n, k, G, keys, oc); b.record(s); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print("route to %d owners: %s ms -> %.1f Gbases/s" % (G, ["%.1f" % t for t in ts], n * L / min(ts) / 1e6))
