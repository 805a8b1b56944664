"""Hash-prefix sharded k-mer counting over torch.distributed (one process per GPU).

The reference partitions its table by `min_mer % n_parts` inside one process and merges
partitions through temp files (counter/src/lib.rs:100,127,188-231).  Here the partitions
are GPUs: rank `o` owns every canonical k-mer with kt_owner_of(kmer, world) == o.

Schedule ("route then count"): each rank turns its own reads into
canonical k-mers grouped by owner (kt_ctr_route, on the GPU), one all-to-all moves each
group to its owner (RCCL over xGMI with backend "nccl" - every peer pair has its own
link, so the exchange uses all 7 links at once; gloo in the CPU tests), and every owner
counts what it received into its local HBM table (kt_ctr_add_pairs).  The result stays
sharded: the union of the ranks' exports is the answer (the reference's output order is
unspecified anyway).  With world == 1 there is no exchange and reads are counted by the
fused kt_ctr_add_reads kernel.

torch is plumbing only: device buffers and the collective.
"""
import torch
import torch.distributed as dist

from . import device


def exchange_keys(keys, send_counts, group=None, recv_buf=None):
    """All-to-all of owner-grouped k-mers.

    keys        1-D int64 tensor; the first sum(send_counts) entries are grouped by owner rank
    send_counts python list / 1-D int64 CPU tensor of length world: group sizes
    recv_buf    optional 1-D int64 tensor on keys.device to receive into when it is large enough (fabric path)
    returns     (received_keys int64 tensor, recv_counts list)
    Works on CUDA tensors (RCCL) and CPU tensors (gloo).
    """
    world = dist.get_world_size(group)
    send = [int(x) for x in send_counts]
    assert len(send) == world
    on_gpu_fabric = dist.get_backend(group) == "nccl"
    # gloo (CPU tests, or several ranks sharing one GPU) moves host memory: stage through it
    xdev = keys.device if on_gpu_fabric else torch.device("cpu")
    s = torch.tensor(send, dtype=torch.int64, device=xdev)
    r = torch.empty(world, dtype=torch.int64, device=xdev)
    dist.all_to_all_single(r, s, group=group)
    recv = [int(x) for x in r.cpu()]
    src = keys[: sum(send)].contiguous().to(xdev)
    if recv_buf is not None and on_gpu_fabric and recv_buf.numel() >= sum(recv):
        out = recv_buf[: sum(recv)]
    else:
        out = torch.empty(sum(recv), dtype=torch.int64, device=xdev)
    dist.all_to_all_single(out, src, output_split_sizes=recv, input_split_sizes=send, group=group)
    return out.to(keys.device), recv


class ShardedCounter:
    """CountComputer state spread over the ranks of `group` (None = single GPU)."""

    def __init__(self, ctx, k, capacity_slots, group=None):
        self.ctx = ctx
        self.k = k
        self.group = group
        self.world = 1 if group is None else dist.get_world_size(group)
        self.rank = 0 if group is None else dist.get_rank(group)
        self.table = device.Counter(ctx, k, capacity_slots)
        self._route_buf = None
        self._owner_counts = None
        self._recv_buf = None   # grow-only receive buffer: one 8 B/k-mer array alive instead of two

    def clear(self):
        self.table.clear()

    def add_reads(self, bases, offsets, n_reads):
        """bases/offsets: device tensors (CSR batch of this rank's reads)"""
        if self.world == 1:
            self.table.add_reads(bases, offsets, n_reads)
            return
        total = bases.numel()
        if self._route_buf is None or self._route_buf.numel() < total:
            self._route_buf = torch.empty(total, dtype=torch.int64, device=bases.device)
            self._owner_counts = torch.empty(64, dtype=torch.int64, device=bases.device)
        self.ctx.route(bases, offsets, n_reads, self.k, self.world, self._route_buf, self._owner_counts)
        # wait for the routing kernels (and for the previous batch's table build, which reads the receive
        # buffer) whatever stream the context runs on, before torch reads the counts / refills the buffer
        self.ctx.sync()
        send = self._owner_counts[: self.world].cpu().tolist()
        recv_keys, _ = exchange_keys(self._route_buf, send, self.group, self._recv_buf)
        if self._recv_buf is None or recv_keys.numel() > self._recv_buf.numel():
            self._recv_buf = recv_keys
        if recv_keys.numel():
            if dist.get_backend(self.group) == "nccl":
                torch.cuda.current_stream().synchronize()   # the collective's output is complete
            self.table.add_pairs(recv_keys, None, recv_keys.numel())

    def size_local(self):
        return self.table.size()

    def size_global(self):
        n = self.size_local()
        if self.world == 1:
            return n
        t = torch.tensor([n], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, group=self.group)
        return int(t.item())

    def export_local(self, sort=True):
        return self.table.export_host(sort)

    def close(self):
        self.table.close()
