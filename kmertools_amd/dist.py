"""Hash-prefix sharded k-mer counting, one process per GPU: a thin caller of the C ABI's kt_sharded_*
(include/kmertools_hip.h; kmertools_amd/csrc/kt_shard.hip has the schedule).

The reference partitions its table by `min_mer % n_parts` inside one process and merges
partitions through temp files (counter/src/lib.rs:100,127,188-231).  Here the partitions
are GPUs: rank `o` owns the k-mers whose hash prefix falls into its interval (kt_sharded_owner_of), every rank runs the
first partition pass over its own reads, and the library exchanges the pass's output regions (librccl over xGMI,
called from C) and counts what each rank owns.  torch.distributed only does what a launcher does: it carries the
128-byte RCCL id from rank 0 to the others, or - with the gloo backend, i.e. the CPU tests and several ranks sharing
one GPU - it IS the transport, through the library's host all-to-all callback.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import device


def host_alltoall(group):
    """kt_alltoall_fn over torch.distributed (gloo): block p of `send` goes to rank p (host memory, equal blocks)"""
    world = dist.get_world_size(group)

    def fn(send_addr, recv_addr, bytes_per_rank):
        n = int(bytes_per_rank) * world
        s = torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * n).from_address(send_addr)))
        r = torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * n).from_address(recv_addr)))
        dist.all_to_all_single(r, s, group=group)
        return 0
    return fn


class ShardedCounter:
    """CountComputer state spread over the ranks of `group` (None = single GPU)."""

    def __init__(self, ctx, k, capacity_slots, group=None, max_batch_bases=1 << 30):
        self.ctx, self.k, self.group = ctx, k, group
        self.world = 1 if group is None else dist.get_world_size(group)
        self.rank = 0 if group is None else dist.get_rank(group)
        self.transport = "none (one rank)"
        self.sharded = None
        if self.world > 1 and dist.get_backend(group) == "nccl":
            self.sharded = self._try_rccl(ctx, k, capacity_slots, max_batch_bases)
            if self.sharded is not None:
                self.transport = "librccl ncclSend/ncclRecv from the C ABI"
            else:
                # the library could not bring up its own communicator on every rank: the exchange goes through the
                # host all-to-all callback over a gloo group instead (slow, staged through host memory - said loudly)
                import sys
                if self.rank == 0:
                    print("kmertools_amd: librccl communicator unavailable on some rank - sharded counting falls back "
                          "to the host all-to-all transport (gloo)", file=sys.stderr)
                # (new_group is collective over the DEFAULT group: fine for the whole world - what bench.py passes - and a
                # caller with a sub-group has to bring its own gloo group instead)
                ranks = dist.get_process_group_ranks(group if group is not None else dist.group.WORLD)
                if len(ranks) != dist.get_world_size():
                    raise RuntimeError("sharded counting over a sub-group needs a working librccl communicator")
                self._host_group = dist.new_group(ranks=ranks, backend="gloo")
                self.transport = "host all-to-all over gloo (fallback: librccl communicator unavailable)"
                self.sharded = device.Sharded(ctx, k, capacity_slots, max_batch_bases, self.world, self.rank,
                                              ("host", host_alltoall(self._host_group)))
        elif self.world > 1:
            self.transport = "host all-to-all over torch.distributed"
            self.sharded = device.Sharded(ctx, k, capacity_slots, max_batch_bases, self.world, self.rank,
                                          ("host", host_alltoall(group)))
        else:
            self.sharded = device.Sharded(ctx, k, capacity_slots, max_batch_bases, 1, 0, None)
        self.table = self.sharded.table

    def _try_rccl(self, ctx, k, capacity_slots, max_batch_bases):
        """the library's own RCCL communicator, or None - decided by all ranks together, so that no rank waits in
        ncclCommInitRank for one that could not even load librccl"""
        group = self.group

        def all_ok(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return bool(t.item())

        # step 1, no peer needed: librccl loads and its symbols resolve; the shard and the exchange buffers are allocated
        try:
            uid = device.Sharded.unique_id()
            sh = device.Sharded(ctx, k, capacity_slots, max_batch_bases, self.world, self.rank, ("rccl", b""), connect=False)
            ok = True
        except Exception:
            uid, sh, ok = None, None, False
        if not all_ok(ok):   # some rank could not: nobody enters ncclCommInitRank
            if sh is not None:
                sh.close()
            return None
        # step 2: rank 0's id reaches the others, every rank joins the communicator
        box = [uid if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        sh._transport = ("rccl", box[0])
        try:
            sh.connect()
            ok = True
        except Exception:
            ok = False
        if not all_ok(ok):
            sh.close()
            return None
        return sh

    def clear(self):
        self.sharded.clear()

    def add_reads(self, bases, offsets, n_reads):
        """collective; bases/offsets: device tensors (CSR batch of this rank's reads, at most max_batch_bases)"""
        self.sharded.add_reads(bases, offsets, n_reads)

    def finalize(self):
        """collective; after the last add_reads, before the shard is read"""
        self.sharded.finalize()

    def size_local(self):
        return self.table.size()

    def size_global(self):
        n = self.size_local()
        if self.world == 1:
            return n
        on_fabric = dist.get_backend(self.group) == "nccl"
        t = torch.tensor([n], dtype=torch.int64, device="cuda" if on_fabric else "cpu")
        dist.all_reduce(t, group=self.group)
        return int(t.item())

    def export_local(self, sort=True):
        return self.table.export_host(sort)

    def close(self):
        self.sharded.close()
