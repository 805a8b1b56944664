"""Randomised parity soak on the GPU box: oligo / ctr / cov / cgr / min against the CPU oracle on ragged reads
with random parameters.  usage: python tools/fuzz_parity.py [seconds] [seed]"""
import os, sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np
from kmertools_amd import device
from oracle import kt_oracle as oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = device.Context(0)
os.environ["KT_BULK_MIN_BASES"] = "0"


def reads(clean=False):
    n = int(rng.integers(1, 400))
    style = rng.integers(0, 4)
    max_len = [40, 300, 3000, 20000][style]
    if style == 3:
        n = min(n, 12)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for _ in range(n):
        L = int(rng.integers(0, max_len))
        s = alpha[rng.integers(0, 4, size=L)].copy()
        if L and not clean:
            m = rng.random(L)
            pn = rng.choice([0.0, 0.002, 0.05])
            s[m < pn] = ord("N")
            s[(m > 0.5) & (m < 0.5 + rng.choice([0, 0.1]))] |= 0x20
            s[(m > 0.9) & (m < 0.902)] = ord("U")
            if rng.random() < 0.3:
                s[(m > 0.95) & (m < 0.953)] = rng.integers(0, 4)
        elif L:
            m = rng.random(L)
            s[(m > 0.5) & (m < 0.6)] |= 0x20
        if L > 50 and rng.random() < 0.1:
            a = int(rng.integers(0, L - 40))
            s[a:a + int(rng.integers(10, 40))] = s[a]      # homopolymer run
        out.append(s.tobytes())
    return out


t_end = time.time() + budget
cases = dict(oligo=0, ctr=0, cov=0, cgr=0, min=0)
while time.time() < t_end:
    seqs = reads()
    bases, offsets = oracle.to_csr(seqs)
    # oligo
    k = int(rng.integers(3, 8)); cm = bool(rng.integers(0, 2)); norm = bool(rng.integers(0, 2))
    got = ctx.oligo_host(bases, offsets, k, count_min=cm, norm=norm)
    assert np.array_equal(got, oracle.oligo_batch(bases, offsets, k, cm, norm)), ("oligo", seed, k, cm, norm)
    if rng.random() < 0.3:
        g32 = ctx.oligo_host(bases, offsets, k, count_min=cm, norm=True, dtype="f32")
        assert np.abs(g32.astype(np.float64) - oracle.oligo_batch(bases, offsets, k, cm, True)).max() <= 1e-6
        gu = ctx.oligo_host(bases, offsets, k, count_min=cm, norm=False, dtype="u32")
        assert np.array_equal(gu.astype(np.float64), oracle.oligo_batch(bases, offsets, k, cm, False))
    cases["oligo"] += 1
    # ctr + cov
    k = int(rng.integers(1, 32))
    wk, wc = oracle.count_reads(bases, offsets, k)
    cap = max(4096, int(len(wk) * rng.choice([2.1, 2.6, 3.3])))   # rounded up inside the library
    ctr = device.Counter(ctx, k, cap)
    half = len(seqs) // 2
    b1, o1 = oracle.to_csr(seqs[:half]); b2, o2 = oracle.to_csr(seqs[half:])
    ctr.add_reads_host(b1, o1); ctr.add_reads_host(b2, o2)
    gk, gc = ctr.export_host()
    assert np.array_equal(gk, wk) and np.array_equal(gc, wc), ("ctr", seed, k)
    if rng.random() < 0.5 and int(offsets[-1]):
        # the routed-keys route: canonical k-mers as an array into an empty table (bulk), then once more (atomics)
        f, r, _ = ctx.kmers_host(bases, offsets, k)
        keys = np.minimum(f, r)
        rng.shuffle(keys)
        ctr.clear()
        ctr.add_pairs_host(keys, None)
        ctr.add_pairs_host(keys[: len(keys) // 3], None)
        uk, uc = np.unique(np.concatenate([keys, keys[: len(keys) // 3]]), return_counts=True)
        gk, gc = ctr.export_host()
        assert np.array_equal(gk, uk) and np.array_equal(gc, uc.astype(np.uint32)), ("pairs", seed, k)
        ctr.clear()
        ctr.add_reads_host(bases, offsets)
    cases["ctr"] += 1
    q = reads()
    qb, qo = oracle.to_csr(q)
    oc = oracle.Counter(1); oc.add_reads(bases, offsets, k)
    bs, bc = int(rng.integers(1, 9)), int(rng.integers(1, 40))
    assert np.array_equal(ctr.cov_host(qb, qo, bs, bc, True), oc.cov_batch(qb, qo, k, bs, bc, True)), ("cov", seed, k, bs, bc)
    cases["cov"] += 1
    ctr.close()
    # cgr
    cs = reads(clean=True)
    cb, co = oracle.to_csr(cs)
    v = int(rng.choice([1, 2, 7, 16, 1000]))
    got = ctx.cgr_host(cb, co, v)
    want = oracle.cgr_batch(cb, co, v) if int(co[-1]) else np.zeros((0, 2))
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), ("cgr", seed, v)
    cases["cgr"] += 1
    # min
    m = int(rng.integers(1, 32)); w = int(rng.choice([0, m, m + 1, m + int(rng.integers(0, 60)), m + 1023, m + 1024, m + 4095]))
    ms = [s for s in seqs if w != 0 or len(s) >= m]
    mb, mo = oracle.to_csr(ms)
    evo, kk, ss, ee = ctx.minimisers_host(mb, mo, w, m)
    for i, s in enumerate(ms):
        g = [(int(kk[j]), int(ss[j]), int(ee[j])) for j in range(int(evo[i]), int(evo[i + 1]))]
        assert g == oracle.minimisers(s, w, m), ("min", seed, i, w, m)
    cases["min"] += 1
    # min, windows of more than 4096 m-mers (the two-level sliding minimum), a few long reads with N runs and repeats
    if rng.random() < 0.3:
        m = int(rng.integers(1, 32))
        w = m - 1 + int(rng.choice([4097, 4096 + int(rng.integers(1, 4096)), 8192, 8193, 3 * 4096 + int(rng.integers(0, 4096)),
                                    int(rng.integers(17_000, 40_000))]))
        ls = []
        for L in (int(rng.integers(0, 2 * w)), int(rng.integers(w, 3 * w)), w, w + 1, int(rng.integers(1, 300))):
            t = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=L)].copy()
            if L > 100 and rng.random() < 0.5:
                at = int(rng.integers(0, L - 50)); t[at:at + int(rng.integers(1, 5))] = ord("N")
            if L > 1000 and rng.random() < 0.3:
                per = int(rng.integers(1, 40)); at = int(rng.integers(0, L // 2)); span = int(rng.integers(1, L // 2))
                t[at:at + span] = np.resize(t[at:at + per].copy(), span)
            ls.append(t.tobytes())
        lb, lo = oracle.to_csr(ls)
        evo, kk, ss, ee = ctx.minimisers_host(lb, lo, w, m)
        for i, s in enumerate(ls):
            g = [(int(kk[j]), int(ss[j]), int(ee[j])) for j in range(int(evo[i]), int(evo[i + 1]))]
            assert g == oracle.minimisers(s, w, m), ("min wide", seed, i, w, m)
        cases["min_wide"] = cases.get("min_wide", 0) + 1
print("fuzz ok", cases, "seed", seed)
