"""world_size-2 CPU (gloo) test of the multi-GPU counting schedule's host side: ownership by the k-mer's minimiser
(kt_shard_minimiser / kt_shard_owner_of - the library's host functions - against tests/shard_ref.py's restatement: the
owners partition the k-mers, both strands of a k-mer have one owner), the wire format (records of at most 8 k-mers of
one owner, 80 bits each, in blocks of 1024), and the host all-to-all transport that kt_sharded_connect_host drives
(kmertools_amd.dist.host_alltoall over torch.distributed: equal blocks, block p to rank p).
The GPU kernels are replaced here by shard_ref.py and the oracle (test infrastructure): every rank cuts its reads into
records, fills one message per owner - a status word, the record count, the blocks, like the library's regions - the
transport moves them, and what arrives is decoded and counted."""
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
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import shard_ref
        canon = _canonical_kmers(oracle, bases, offsets, K)
        assert device.shard_minimiser(K) == (shard_ref.mmer_of(K), shard_ref.window_of(K)) == (16, 16)
        for kk in (1, 7, 8, 9, 10, 11, 14, 15, 16, 21, 22, 23, 31):     # m >= 8 wherever k allows, w a power of two <= 16
            m, w = device.shard_minimiser(kk)
            assert (m, w) == (shard_ref.mmer_of(kk), shard_ref.window_of(kk)) and m + w - 1 == kk
            assert w in (1, 2, 4, 8, 16) and (m >= 8 or w == 1) and m <= 16
        # the library's host function against the restatement, both strands, several rank counts and k
        for kk, nr in ((K, WORLD), (K, 8), (21, 3), (15, 5), (10, 2), (6, 4)):
            sub = canon[:400] & np.uint64((1 << (2 * kk)) - 1)
            ref = shard_ref.owner_of_kmers(sub, kk, nr)
            got = np.array([device.shard_owner_of(int(x), kk, nr) for x in sub], dtype=np.int64)
            rc = np.array([device.shard_owner_of(oracle.rev_comp(int(x), kk), kk, nr) for x in sub], dtype=np.int64)
            assert np.array_equal(ref, got) and np.array_equal(got, rc) and got.min() >= 0 and got.max() < nr
        owners = shard_ref.owner_of_kmers(canon, K, WORLD)
        assert 0.3 < (owners == 0).mean() < 0.7                         # (balanced: the order is a hash of the m-mer)
        # the records of this rank's reads, one stream per owner
        streams = [[] for _ in range(WORLD)]
        n_kmers = 0
        for i in range(len(offsets) - 1):
            codes, valid = shard_ref.codes_of(bases[int(offsets[i]):int(offsets[i + 1])].tobytes())
            for o, start, n in shard_ref.records_of_read(codes, valid, K, WORLD):
                assert 1 <= n <= shard_ref.REC_KMERS
                streams[o].append(shard_ref.pack_record(codes, start, n, K))
                n_kmers += n
        assert n_kmers == len(canon)                                    # every k-mer is in exactly one record
        assert len(canon) / sum(len(x) for x in streams) > 4.5          # (and the records are mostly full: ~6 k-mers each)
        # one message per owner: [status, records, blocks...], equal sizes (the host transport's contract)
        max_blocks = (N_PER_RANK * (L - K + 1) + shard_ref.BLOCK_RECS - 1) // shard_ref.BLOCK_RECS
        words = 8 + max_blocks * shard_ref.BLOCK_WORDS
        msg_bytes = words * 8
        send = np.zeros(WORLD * words, np.uint64)
        for o in range(WORLD):
            blk = shard_ref.blocks_of_records(streams[o])
            send[o * words + 1] = len(streams[o])
            send[o * words + 8:o * words + 8 + len(blk)] = blk
        recv = np.full(WORLD * words, 0xDEAD, np.uint64)
        fn = ktdist.host_alltoall(dist.group.WORLD)          # what kt_sharded_connect_host is given
        assert fn(send.ctypes.data, recv.ctypes.data, msg_bytes) == 0
        got = []
        for p in range(WORLD):
            assert int(recv[p * words]) == 0                  # the sender's status word
            n = int(recv[p * words + 1])
            for a, b in shard_ref.records_of_blocks(recv[p * words + 8:(p + 1) * words], n):
                got.extend(shard_ref.kmers_of_record(a, b, K))
        got = np.array(got, dtype=np.uint64)
        # everything received is owned by this rank
        assert all(device.shard_owner_of(int(x), K, WORLD) == rank for x in got[:500])
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
