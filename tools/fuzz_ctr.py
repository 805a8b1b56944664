"""Soak of the round-2 counting paths against the CPU oracle: range build at random load factors, in-place rebuild
merges and probing inserts in random order, the dense table state, hash-partition passes (kt_ctr_add_reads_part),
the routed (sharded) path with one rank through RCCL, lookups (cov) on whatever layout results.
usage: python tools/fuzz_ctr.py [seconds] [seed]"""
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


def batch(total):
    style = rng.integers(0, 3)
    if style == 0:
        lens = np.full(max(total // 150, 1), 150)
    elif style == 1:
        lens = rng.integers(0, 600, size=max(total // 300, 1))
    else:
        lens = np.concatenate([rng.integers(1000, max(total // 4, 1001), size=4), rng.integers(0, 50, size=500)])
        rng.shuffle(lens)
    offsets = np.zeros(len(lens) + 1, np.uint64)
    offsets[1:] = np.cumsum(lens)
    n = int(offsets[-1])
    b = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=n)].copy()
    if rng.random() < 0.3:
        period = int(rng.integers(1, 60))
        span = int(n * rng.choice([0.02, 0.2]))
        at = int(rng.integers(0, n - span + 1))
        b[at:at + span] = np.resize(b[at:at + period].copy(), span)
    m = rng.random(n)
    b[(m > 0.3) & (m < 0.35)] |= 0x20
    b[m < rng.choice([0.0, 0.0005, 0.01])] = ord("N")
    return b, offsets


def dev(a, dt):
    return torch.from_numpy(a.view(dt) if a.dtype != dt else a).cuda()


t_end = time.time() + budget
rounds = 0
while time.time() < t_end:
    k = int(rng.choice([9, 13, 15, 16, 17, 21, 31]))
    nb = int(rng.integers(1, 5))
    batches = [batch(int(rng.integers(100_000, 1_500_000))) for _ in range(nb)]
    oc = oracle.Counter(4)
    for hb, ho in batches:
        oc.add_reads(hb, ho, k, threads=8)
    wk, wc = oc.export()
    mode = rng.choice(["table", "parts", "sharded"])
    os.environ["KT_BULK_MIN_BASES"] = "0"
    os.environ["KT_BULK_DENSE"] = str(int(rng.integers(0, 2)))
    os.environ["KT_BULK_MERGE_DIV"] = str(int(rng.choice([8, 10 ** 9])))
    factor = float(rng.choice([1.15, 1.3, 1.6, 2.1, 3.4]))
    cap = max(1 << 15, int(len(wk) * factor))
    tag = (seed, rounds, k, mode, nb, factor, os.environ["KT_BULK_DENSE"], os.environ["KT_BULK_MERGE_DIV"])
    if mode == "parts":
        P = int(rng.integers(2, 6))
        ctr = device.Counter(ctx, k, max(1 << 15, int(cap / P * 1.3)))
        gk, gc = [], []
        for p in range(P):
            ctr.clear()
            for hb, ho in batches:
                os.environ["KT_BULK"] = str(int(rng.integers(0, 2)))
                ctr.add_reads_part(dev(hb, np.uint8), dev(ho, np.int64), len(ho) - 1, P, p)
            a, c = ctr.export_host(sort=False)
            gk.append(a); gc.append(c)
        os.environ["KT_BULK"] = "1"
        gk, gc = np.concatenate(gk), np.concatenate(gc)
        o = np.argsort(gk)
        assert np.array_equal(gk[o], wk) and np.array_equal(gc[o], wc), ("parts",) + tag
        ctr.close()
    else:
        if mode == "sharded":
            os.environ["KT_SHARD_FORCE"] = str(int(rng.integers(1, 10)))   # a single rank routing into 1 .. 9 owners' regions
            os.environ["KT_SHARD_SLICES"] = str(int(rng.integers(1, 6)))
            sh = device.Sharded(ctx, k, cap, max(int(ho[-1]) for _, ho in batches) + 1, 1, 0, None)
            os.environ["KT_SHARD_FORCE"] = "0"
            for hb, ho in batches:
                sh.add_reads(dev(hb, np.uint8), dev(ho, np.int64), len(ho) - 1)
            sh.finalize()
            ctr = sh.table
        else:
            sh = None
            ctr = device.Counter(ctx, k, cap)
            for hb, ho in batches:
                os.environ["KT_BULK"] = str(int(rng.integers(0, 2)))
                ctr.add_reads(dev(hb, np.uint8), dev(ho, np.int64), len(ho) - 1)
            os.environ["KT_BULK"] = "1"
        try:
            d = ctr.size()
        except Exception as e:   # a range may genuinely be full at the tightest factors: that must be the loud error
            assert "full" in str(e) and factor <= 1.3, ("unexpected error", str(e)) + tag
            (sh or ctr).close()
            rounds += 1
            continue
        assert d == len(wk), ("size", d, len(wk)) + tag
        if rng.random() < 0.5:   # lookups first: a dense table gets its probing image here
            hb, ho = batches[0]
            bs, bc = int(rng.integers(1, 5)), int(rng.integers(2, 20))
            cov = torch.empty((len(ho) - 1, bc), dtype=torch.float64, device="cuda")
            ctr.cov(dev(hb, np.uint8), dev(ho, np.int64), len(ho) - 1, bs, bc, cov, norm=True)
            ctx.sync()
            assert np.array_equal(cov.cpu().numpy(), oc.cov_batch(hb, ho, k, bs, bc, True)), ("cov",) + tag
        gk, gc = ctr.export_host()
        assert np.array_equal(gk, wk) and np.array_equal(gc, wc), ("export",) + tag
        (sh or ctr).close()
    rounds += 1
print("fuzz_ctr ok rounds", rounds, "seed", seed)
