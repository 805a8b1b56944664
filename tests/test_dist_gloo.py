"""world_size-2 CPU (gloo) test of the multi-GPU counting schedule's host side: ownership by hash prefix
(kt_shard_layout / kt_shard_owner_of: every rank derives the same layout, the owners partition the k-mers, a rank's
k-mers have hash prefixes inside its bucket interval) and the host all-to-all transport that kt_sharded_connect_host
drives (kmertools_amd.dist.host_alltoall over torch.distributed: equal blocks, block p to rank p).
The GPU kernels are replaced here by the oracle (test infrastructure): every rank fills one block per owner - a status
word, the key count, the keys, like the library's blocks - the transport moves them, and what arrives is counted."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2
K = 31
N_PER_RANK = 300
L = 150
SEED = 99


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _canonical_kmers(oracle, bases, offsets, k):
    out = []
    for i in range(len(offsets) - 1):
        f, r, _ = oracle.kmers(bases[int(offsets[i]):int(offsets[i + 1])].tobytes(), k)
        out.append(np.minimum(f, r))
    return np.concatenate(out) if out else np.zeros(0, np.uint64)


def _worker(rank, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from kmertools_amd import _lib, device, dist as ktdist
        from oracle import kt_oracle as oracle
        bases, offsets = oracle.synth_reads(SEED, N_PER_RANK, L, noise=True, genome_len=20000,
                                            first_read=rank * N_PER_RANK)
        canon = _canonical_kmers(oracle, bases, offsets, K)
        cap = 1 << 20
        bits, lo, hi, slots = device.shard_layout(cap, WORLD, rank)
        layouts = [device.shard_layout(cap, WORLD, r) for r in range(WORLD)]
        assert all(x[0] == bits for x in layouts) and slots >= cap          # every rank derives the same prefix bits
        assert layouts[0][1] == 0 and layouts[-1][2] == 1 << bits           # the intervals tile the buckets
        assert all(layouts[r][2] == layouts[r + 1][1] for r in range(WORLD - 1))
        owners = np.array([device.shard_owner_of(int(x), bits, WORLD) for x in canon], dtype=np.int64)
        # one block per owner: [status, count, keys...], equal sizes (the host transport's contract)
        words = 8 + N_PER_RANK * (L - K + 1)                  # (the same on every rank: at most this many k-mers)
        msg_bytes = words * 8
        send = np.zeros(WORLD * words, np.uint64)
        for o in range(WORLD):
            mine = canon[owners == o]
            send[o * words + 1] = len(mine)
            send[o * words + 8:o * words + 8 + len(mine)] = mine
        recv = np.full(WORLD * words, 0xDEAD, np.uint64)
        fn = ktdist.host_alltoall(dist.group.WORLD)          # what kt_sharded_connect_host is given
        assert fn(send.ctypes.data, recv.ctypes.data, msg_bytes) == 0
        got = []
        for p in range(WORLD):
            assert int(recv[p * words]) == 0                  # the sender's status word
            n = int(recv[p * words + 1])
            got.append(recv[p * words + 8:p * words + 8 + n])
        got = np.concatenate(got)
        # everything received is owned by this rank
        assert all(device.shard_owner_of(int(x), bits, WORLD) == rank for x in got[:500])
        ctr = oracle.Counter(1)
        ctr.add_pairs(got, np.ones(len(got), np.uint32))
        k_, c_ = ctr.export()
        q.put((rank, k_, c_, int(len(canon))))
    finally:
        dist.destroy_process_group()


def test_two_rank_route_exchange_count(oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(WORLD)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort(key=lambda x: x[0])
    keys = np.concatenate([r[1] for r in results])
    counts = np.concatenate([r[2] for r in results])
    # shards are disjoint
    assert len(np.unique(keys)) == len(keys)
    order = np.argsort(keys)
    keys, counts = keys[order], counts[order]
    bases, offsets = oracle.synth_reads(SEED, WORLD * N_PER_RANK, L, noise=True, genome_len=20000)
    wk, wc = oracle.count_reads(bases, offsets, K)
    assert np.array_equal(keys, wk) and np.array_equal(counts, wc)
    assert int(counts.sum()) == sum(r[3] for r in results)
    assert wc.max() > 1
