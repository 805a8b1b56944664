"""Soak of the sharded counter with several ranks on ONE GPU (host all-to-all over gloo): every iteration a new counter -
random k (every minimiser window: w = 16, 8, 4, 2, 1), table size, pieces per exchange, regions small enough to overflow or
not, batches below or above the partition passes' size - and one to three collective batches whose sizes differ from rank to
rank (empty ones among them), reads of random lengths, sometimes a k-mer that floods its owner; the union of the shards must be
the CPU oracle's table of all the reads, and every k-mer must lie on the rank its minimiser gives it.  Every rank derives the iteration's shape from the same seed.
usage: python tools/fuzz_shard.py [seconds] [seed] [ranks]"""
import os, sys, time, pathlib, socket
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np


def worker(rank, world, port, budget, seed, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["KT_BULK_MIN_BASES"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kmertools_amd import device, dist as ktdist
    from oracle import kt_oracle as oracle
    torch.cuda.set_device(0)
    ctx = device.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    t0 = time.time()
    it = 0
    try:
      try:
        while True:
            # (rank 0 decides when to stop: every rank must leave in the same iteration)
            go = [time.time() - t0 < budget]
            dist.broadcast_object_list(go, src=0)
            if not go[0]:
                break
            rng = np.random.default_rng([seed, it])          # the same on every rank
            k = int(rng.choice([7, 9, 11, 15, 16, 17, 21, 22, 23, 27, 31]))
            slices = int(rng.integers(1, 6))
            presplit = bool(rng.integers(0, 2))   # (here: small regions, so that floods and plain batches overflow them)
            os.environ["KT_SHARD_SLICES"] = str(slices)
            os.environ["KT_BULK_MIN_BASES"] = "0" if rng.integers(0, 4) else str(1 << 40)
            n_batches = int(rng.integers(1, 4))
            max_bases = 1 << int(rng.integers(18, 22))
            if presplit:   # regions that ordinary reads fit (0.2 records per base and owner's share) and a flood does not
                os.environ["KT_SHARD_ROOM_BLOCKS"] = os.environ.get("KT_FUZZ_ROOM_BLOCKS", str(int(0.2 * max_bases / world / 1024) + 4))
            else:
                os.environ.pop("KT_SHARD_ROOM_BLOCKS", None)
            cap = max(1 << 21, int(2.2 * world * n_batches * max_bases))   # slots: every base a distinct k-mer would still fit
            sc = ktdist.ShardedCounter(ctx, k, cap, group=dist.group.WORLD, max_batch_bases=max_bases)
            mine_b, mine_o = [], []
            for b in range(n_batches):
                sizes = rng.integers(0, max_bases, size=world) * (rng.integers(0, 4, size=world) > 0)   # bases per rank, some 0
                style = int(rng.integers(0, 3))
                flood = rng.integers(0, 4, size=world) == 0
                r2 = np.random.default_rng([seed, it, b, rank])
                total = int(sizes[rank])
                if style == 0:
                    lens = np.full(total // 150, 150)
                elif style == 1:
                    lens = r2.integers(0, 400, size=max(total // 200, 0))
                else:
                    lens = np.concatenate([r2.integers(1000, max(total // 3, 1001), size=2), r2.integers(0, 40, size=50)]) if total > 4000 else np.zeros(0, np.int64)
                    r2.shuffle(lens)
                while lens.sum() > max_bases:
                    lens = lens[: len(lens) // 2]
                offs = np.zeros(len(lens) + 1, np.uint64)
                offs[1:] = np.cumsum(lens)
                n = int(offs[-1])
                hb = np.frombuffer(b"ACGT", np.uint8)[r2.integers(0, 4, size=n)].copy()
                if n:
                    hb[r2.integers(0, n, size=n // 500)] = ord("N")
                if flood[rank] and n > 1000:
                    hb[: n // 3] = ord("A")
                db = torch.from_numpy(hb).cuda() if n else torch.empty(0, dtype=torch.uint8, device="cuda")
                do = torch.from_numpy(offs.astype(np.int64)).cuda()
                sc.add_reads(db, do, len(lens))
                mine_b.append(hb)
                mine_o.append(offs)
            sc.finalize()
            keys, counts = sc.export_local()
            total = sc.size_global()
            strays = sum(1 for x in keys[:200] if sc.sharded.owner_of(int(x)) != rank)
            sc.close()
            if strays:
                q.put("STRAY k-mers on rank %d it=%d k=%d" % (rank, it, k))
                os._exit(1)
            gathered = [None] * world
            dist.all_gather_object(gathered, (keys, counts, mine_b, mine_o))
            if rank == 0:
                ctr = oracle.Counter(4)
                for _, _, bs, os_ in gathered:
                    for hb, offs in zip(bs, os_):
                        if len(offs) > 1:
                            ctr.add_reads(hb, offs, k, threads=4)
                wk, wc = ctr.export()
                gk = np.concatenate([g[0] for g in gathered])
                gc = np.concatenate([g[1] for g in gathered])
                order = np.argsort(gk)
                ok = total == len(wk) and np.array_equal(gk[order], wk) and np.array_equal(gc[order], wc)
                if not ok:
                    q.put("MISMATCH it=%d k=%d slices=%d small_regions=%s batches=%d got %d want %d" % (it, k, slices, presplit, n_batches, len(gk), len(wk)))
                    os._exit(1)
            it += 1
        if rank == 0:
            q.put("fuzz_shard ok iterations %d seed %d ranks %d" % (it, seed, world))
      except BaseException as e:  # noqa: BLE001 - say what happened before the process goes (the parent waits on the queue)
        import traceback
        q.put("rank %d it=%d: %s" % (rank, it, "".join(traceback.format_exception_only(type(e), e)).strip()[-400:]))
        raise
    finally:
        ctx.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=worker, args=(r, world, port, budget, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        print(q.get(timeout=budget + 180))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    sys.exit(0 if all(p.exitcode == 0 for p in procs) else 1)
