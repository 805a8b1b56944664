"""Windows of more than 4096 m-mers: the two-level sliding minimum in front of the tile kernel against the iterator run one
read per thread (KT_MIN_SERIAL=1), on a few long reads.  usage (GPU box): python3 tools/min_wide_timing.py [bases per read] [reads] [w] [m]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from kmertools_amd import device  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
m = int(sys.argv[4]) if len(sys.argv) > 4 else 15
ctx = device.Context(0, torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(7)
bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[torch.randint(0, 4, (n * L,), device="cuda", generator=g)]
offsets = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device="cuda")
evo = torch.empty(n + 1, dtype=torch.int64, device="cuda")
res = {}
for serial in ("0", "1"):
    os.environ["KT_MIN_SERIAL"] = serial
    dummy = torch.empty(1, dtype=torch.int64, device="cuda")
    cnt = ctx.minimisers(bases, offsets, n, w, m, evo, dummy, dummy, dummy, 0)
    k = torch.empty(cnt, dtype=torch.int64, device="cuda")
    s = torch.empty(cnt, dtype=torch.int64, device="cuda")
    e = torch.empty(cnt, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert ctx.minimisers(bases, offsets, n, w, m, evo, k, s, e, cnt) == cnt
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[serial] = (cnt, k.clone(), s.clone(), e.clone())
    print("%s: %d reads x %d bases, w=%d m=%d: %d minimisers, %.1f ms (%.2f Gbases/s)" % (
        "iterator, one read per thread" if serial == "1" else "two-level sliding minimum", n, L, w, m, cnt, dt * 1e3, n * L / dt / 1e9))
a, b = res["0"], res["1"]
print("identical:", a[0] == b[0] and all(bool((x == y).all()) for x, y in zip(a[1:], b[1:])))
