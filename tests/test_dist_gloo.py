"""world_size-2 CPU (gloo) test of the multi-GPU counting schedule's host side: ownership by hash prefix, the layout
of the exchanged regions (kt_sharded_message_bytes: a 64-byte header with the key count, then the keys) and the host
all-to-all transport that kt_sharded_create_host drives (kmertools_amd.dist.host_alltoall over torch.distributed).
The GPU kernels are replaced here by the oracle (test infrastructure): every rank fills its regions the way
route_regions_kernel does, the transport moves them, and the received regions are counted."""
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
        owners = np.array([device.owner_of(int(x), WORLD) for x in canon], dtype=np.int64)
        # one slice: WORLD regions of the size the library would use for a batch of this many bases
        msg_bytes = int(_lib.lib().kt_sharded_message_bytes(N_PER_RANK * L, WORLD, 1))
        words = msg_bytes // 8
        assert msg_bytes % 8 == 0 and words - 8 >= (canon.size // WORLD)
        send = np.zeros(WORLD * words, np.uint64)
        for o in range(WORLD):
            mine = canon[owners == o]
            send[o * words] = len(mine)                       # header: key count
            send[o * words + 8:o * words + 8 + len(mine)] = mine
        recv = np.full(WORLD * words, 0xDEAD, np.uint64)
        fn = ktdist.host_alltoall(dist.group.WORLD)          # what kt_sharded_create_host is given
        assert fn(send.ctypes.data, recv.ctypes.data, msg_bytes) == 0
        got = []
        for p in range(WORLD):
            n = int(recv[p * words])
            got.append(recv[p * words + 8:p * words + 8 + n])
        got = np.concatenate(got)
        # everything received is owned by this rank
        assert all(device.owner_of(int(x), WORLD) == rank for x in got[:500])
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
