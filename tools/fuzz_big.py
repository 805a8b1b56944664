"""Parity soak on mid-size batches (millions of bases, device-memory API): every kernel against the CPU oracle.
usage: python tools/fuzz_big.py [seconds] [seed]"""
import os, sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from kmertools_amd import device
from oracle import kt_oracle as oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = device.Context(0, stream=torch.cuda.current_stream().cuda_stream)


def batch(clean=False):
    total = int(rng.integers(200_000, 3_000_000))
    style = rng.integers(0, 3)
    if style == 0:
        lens = np.full(total // 150, 150)
    elif style == 1:
        lens = rng.integers(0, 600, size=total // 300)
    else:
        lens = np.concatenate([rng.integers(1000, 400_000, size=6), rng.integers(0, 50, size=2000)])
        rng.shuffle(lens)
    offsets = np.zeros(len(lens) + 1, np.uint64)
    offsets[1:] = np.cumsum(lens)
    n = int(offsets[-1])
    b = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=n)].copy()
    if rng.random() < 0.3:   # heavy hitters: a tandem repeat over part of the batch (spill list / exact-offset redo)
        period = int(rng.integers(1, 60))
        span = int(n * rng.choice([0.02, 0.2, 0.9]))
        at = int(rng.integers(0, n - span + 1))
        b[at:at + span] = np.resize(b[at:at + period].copy(), span)
    m = rng.random(n)
    b[(m > 0.3) & (m < 0.35)] |= 0x20
    if not clean:
        b[m < rng.choice([0.0, 0.0005, 0.01])] = ord("N")
        b[(m > 0.9) & (m < 0.9005)] = ord("U")
    return b, offsets


def dev(a, dt):
    return torch.from_numpy(a.view(dt) if a.dtype != dt else a).cuda()


t_end = time.time() + budget
rounds = 0
while time.time() < t_end:
    hb, ho = batch()
    n = len(ho) - 1
    db, do = dev(hb, np.uint8), dev(ho, np.int64)
    # oligo
    k = int(rng.integers(3, 7)); cm = bool(rng.integers(0, 2))
    bins = device.bins(k, cm)
    out = torch.empty((n, bins), dtype=torch.float64, device="cuda")
    ctx.oligo(db, do, n, k, out, count_min=cm, norm=True)
    ctx.sync()
    assert np.array_equal(out.cpu().numpy(), oracle.oligo_batch(hb, ho, k, cm, True, threads=8)), ("oligo", seed, rounds)
    # ctr (bulk build on every other round, atomics otherwise) + cov
    os.environ["KT_BULK_MIN_BASES"] = "0" if rounds % 2 == 0 else str(1 << 40)
    k = int(rng.choice([9, 15, 21, 31]))
    wk, wc = oracle.count_reads(hb, ho, k, n_parts=8, threads=8)
    ctr = device.Counter(ctx, k, max(1 << 14, int(len(wk) * rng.choice([2.1, 2.7, 3.4]))))   # rounded up inside the library
    ctr.add_reads(db, do, n)
    d = ctr.size()
    assert d == len(wk), ("ctr size", seed, rounds, k)
    keys = torch.empty(d, dtype=torch.int64, device="cuda"); cnt = torch.empty(d, dtype=torch.int32, device="cuda")
    ctr.export(keys, cnt, d)
    gk = keys.cpu().numpy().view(np.uint64); gc = cnt.cpu().numpy().view(np.uint32)
    o = np.argsort(gk)
    assert np.array_equal(gk[o], wk) and np.array_equal(gc[o], wc), ("ctr", seed, rounds, k)
    bs, bc = int(rng.integers(1, 5)), int(rng.integers(2, 20))
    cov = torch.empty((n, bc), dtype=torch.float64, device="cuda")
    ctr.cov(db, do, n, bs, bc, cov, norm=True)
    ctx.sync()
    oc = oracle.Counter(1); oc.add_reads(hb, ho, k)
    assert np.array_equal(cov.cpu().numpy(), oc.cov_batch(hb, ho, k, bs, bc, True)), ("cov", seed, rounds, k)
    ctr.close()
    # min
    m = int(rng.integers(3, 32)); w = m + int(rng.choice([0, 1, 24, 200, 1000, 1500, 4000]))
    evo = torch.empty(n + 1, dtype=torch.int64, device="cuda"); one = torch.empty(1, dtype=torch.int64, device="cuda")
    ne = ctx.minimisers(db, do, n, w, m, evo, one, one, one, 0)
    mk = torch.empty(max(ne, 1), dtype=torch.int64, device="cuda"); ms = torch.empty_like(mk); me = torch.empty_like(mk)
    assert ctx.minimisers(db, do, n, w, m, evo, mk, ms, me, max(ne, 1)) == ne
    he = evo.cpu().numpy(); hk = mk.cpu().numpy().view(np.uint64); hs = ms.cpu().numpy(); hen = me.cpu().numpy()
    raw = hb.tobytes()
    pick = rng.choice(n, size=min(n, 400), replace=False)
    for i in pick:
        s = raw[int(ho[i]):int(ho[i + 1])]
        if len(s) > 4000:
            continue
        g = [(int(hk[j]), int(hs[j]), int(hen[j])) for j in range(int(he[i]), int(he[i + 1]))]
        assert g == oracle.minimisers(s, w, m), ("min", seed, rounds, i, w, m)
    want_total = sum(len(oracle.minimisers(raw[int(ho[i]):int(ho[i + 1])], w, m)) for i in range(n)) if n < 3000 else None
    if want_total is not None:
        assert ne == want_total, ("min total", seed, rounds, w, m)
    # cgr
    cb, co = batch(clean=True)
    dcb, dco = dev(cb, np.uint8), dev(co, np.int64)
    xy = torch.empty((len(cb), 2), dtype=torch.float64, device="cuda"); bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    v = int(rng.choice([1, 3, 16]))
    ctx.cgr(dcb, dco, len(co) - 1, v, xy, bad)
    ctx.sync()
    assert int(bad.item()) == -1
    assert np.array_equal(xy.cpu().numpy().view(np.uint64), oracle.cgr_batch(cb, co, v).view(np.uint64)), ("cgr", seed, rounds)
    rounds += 1
print("fuzz_big ok rounds", rounds, "seed", seed)
