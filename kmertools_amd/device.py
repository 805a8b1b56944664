"""Thin object layer over the C ABI: a Context (one GPU + one stream) and a Counter
(HBM-resident k-mer table).  Two calling styles, same entry points:

* host arrays (numpy): the library stages them through its own device scratch
  (KT_MEM_HOST) - what the `pykmertools` mirror uses;
* device tensors (anything with `.data_ptr()`, e.g. torch tensors on the GPU):
  pointers are passed straight through (KT_MEM_DEVICE) and work is enqueued on the
  context's stream without synchronising - what bench.py and the parity tests use.
  torch is only plumbing here (allocation, streams, torch.distributed).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KT_F32, KT_F64, KT_MEM_DEVICE, KT_MEM_HOST, KT_U32, check

_DT = {"f64": KT_F64, "f32": KT_F32, "u32": KT_U32, np.float64: KT_F64, np.float32: KT_F32, np.uint32: KT_U32}
_NP = {KT_F64: np.float64, KT_F32: np.float32, KT_U32: np.uint32}


def _ptr(x):
    """device/host address of a numpy array, a tensor with data_ptr(), an int or None"""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    return C.c_void_p(x.data_ptr())


def bins(k, count_min=True):
    out = C.c_uint64()
    check(_lib.lib().kt_bins(k, int(bool(count_min)), C.byref(out)))
    return out.value


def pos_map(k):
    """(min_mer_pos_map u32[4^k], pos_min_mer u64[kcount], kcount) - kmer.rs:54-73"""
    n = 4 ** k
    m = np.zeros(n, np.uint32)
    pk = np.zeros(n, np.uint64)
    cnt = C.c_uint32()
    check(_lib.lib().kt_pos_map(k, _ptr(m), _ptr(pk), C.byref(cnt)))
    return m, pk[: cnt.value].copy(), cnt.value


def rev_comp(kmer, k):
    return int(_lib.lib().kt_rev_comp(kmer, k))


def numeric_to_kmer(kmer, k):
    buf = C.create_string_buffer(k + 1)
    check(_lib.lib().kt_numeric_to_kmer(kmer, k, buf))
    return buf.value.decode()


def kmer_to_numeric(s):
    b = s.encode("latin-1") if isinstance(s, str) else bytes(s)
    f, r = C.c_uint64(), C.c_uint64()
    check(_lib.lib().kt_kmer_to_numeric(b, len(b), C.byref(f), C.byref(r)))
    return f.value, r.value


def cgr_coords(k, vecsize):
    xy = np.zeros((bins(k, True), 2), np.float64)
    check(_lib.lib().kt_cgr_coords(k, float(vecsize), _ptr(xy)))
    return xy


def owner_of(kmer, n_owners):
    """hash partition of a k-mer among n_owners (out-of-core passes, kt_ctr_route): the LOW hash bits"""
    return int(_lib.lib().kt_owner_of(int(kmer), int(n_owners)))


def shard_minimiser(k):
    """-> (m, w): the minimiser length and the window (m-mers per k-mer) the sharded counter uses for k-mers of length k"""
    m, w = C.c_uint32(), C.c_uint32()
    check(_lib.lib().kt_shard_minimiser(int(k), C.byref(m), C.byref(w)))
    return m.value, w.value


def shard_owner_of(kmer, k, n_ranks):
    """the rank that owns a k-mer (either strand) of length k among n_ranks: by its minimiser's hash (kt_shard_owner_of)"""
    return int(_lib.lib().kt_shard_owner_of(int(kmer), int(k), int(n_ranks)))


def to_csr(seqs):
    """list[str|bytes] -> (bases u8[total], offsets u64[n+1])"""
    bs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    joined = b"".join(bs)
    bases = np.frombuffer(joined, dtype=np.uint8).copy() if joined else np.zeros(0, np.uint8)
    return bases, offsets


def view_tensor(addr, shape, dtype, owner=None):
    """a torch tensor over device memory that somebody else owns (no copy, through __cuda_array_interface__); `owner` is
    kept alive by the tensor"""
    import torch
    typestr = {torch.float64: "<f8", torch.float32: "<f4", torch.uint8: "|u1", torch.int32: "<i4", torch.int64: "<i8"}[dtype]

    class _View:
        pass
    v = _View()
    v.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": typestr, "data": (int(addr), False),
                                  "version": 2, "strides": None}
    v._owner = owner
    t = torch.as_tensor(v, device="cuda")
    t._kt_owner = owner
    return t


class DeviceArray:
    """Device memory handed out by the library (kt_device_alloc_placed), released by close() / garbage collection.
    `tensor(shape, dtype)` views it as a torch tensor (no copy; the view keeps this object alive); `ptr` is the raw
    address for the C-ABI wrappers here, which accept an int where they accept a tensor."""

    def __init__(self, ctx, ptr, nbytes):
        self._ctx, self.ptr, self.nbytes = ctx, int(ptr), int(nbytes)

    def tensor(self, shape, dtype):
        return view_tensor(self.ptr, shape, dtype, owner=self)

    def close(self):
        if self.ptr and self._ctx._h.value:
            _lib.lib().kt_device_free(self._ctx._h, C.c_void_p(self.ptr))
        self.ptr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One device + one stream (kt_ctx).  `stream` is a raw hipStream_t handle
    (e.g. torch.cuda.current_stream().cuda_stream; 0 = the default stream) to enqueue on,
    or None to let the context own a private stream."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        self.device = device
        own = stream is None
        check(_lib.lib().kt_ctx_create(device, None if own else C.c_void_p(int(stream)), int(own),
                                       C.byref(self._h)))

    def alloc_placed(self, nbytes, probe=None, candidates=8, launches=4):
        """kt_device_alloc_placed: a device array of nbytes, the fastest of up to `candidates` allocations under
        probe(address) - a callable that enqueues the work the array is meant for on this context's stream (None: a plain
        allocation).  -> (DeviceArray, {"candidates": n, "ms": [per candidate], "picked": index}); candidate 0 is the
        plain allocation.  An exception in the probe is re-raised here."""
        exc = []

        def cb(_user, ptr):
            try:
                probe(int(ptr))
                return 0
            except BaseException as e:  # noqa: BLE001 (ctypes would swallow it)
                exc.append(e)
                return 1
        fn = _lib.PROBE_FN(cb) if probe is not None else None
        out = C.c_void_p()
        ms = (C.c_double * 16)()
        n, picked = C.c_int(0), C.c_int(0)
        rc = _lib.lib().kt_device_alloc_placed(self._h, int(nbytes), int(candidates), int(launches),
                                               C.cast(fn, C.c_void_p) if fn is not None else None, None, C.byref(out), ms,
                                               C.byref(n), C.byref(picked))
        if exc:
            raise exc[0]
        check(rc)
        info = {"candidates": n.value, "ms": [round(ms[i], 4) for i in range(n.value)], "picked": picked.value}
        return DeviceArray(self, out.value, nbytes), info

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().kt_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(_lib.lib().kt_ctx_sync(self._h))

    # -- comp oligo / comp cgr -k ------------------------------------------------------
    def oligo(self, bases, offsets, n_reads, k, out, count_min=True, norm=True, total_step=1, dtype="f64",
              mem=KT_MEM_DEVICE):
        check(_lib.lib().kt_oligo_batch(self._h, _ptr(bases), _ptr(offsets), n_reads, k, int(bool(count_min)),
                                        int(bool(norm)), int(total_step), _DT[dtype], _ptr(out), mem))
        return out

    def oligo_host(self, bases, offsets, k, count_min=True, norm=True, total_step=1, dtype="f64"):
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = len(offsets) - 1
        out = np.zeros((n, bins(k, count_min)), _NP[_DT[dtype]])
        if n:
            self.oligo(bases if bases.size else np.zeros(1, np.uint8), offsets, n, k, out, count_min, norm,
                       total_step, dtype, KT_MEM_HOST)
        return out

    def oligo_tuning(self, mode):
        """False / 0: the launches that follow take no part in the k = 4 launch-shape measurement (a caller timing
        launches itself); True / 1: resume; n >= 2: as 0, with n workgroups per resident slot for every launch"""
        check(_lib.lib().kt_oligo_tuning(self._h, int(mode)))

    def oligo_launch_info(self):
        """-> dict: workgroups per resident slot of the k = 4 launches into the latest output array, whether that has
        been measured yet, and the measured ns per read of the candidates 32 / 96 / 200"""
        w, d, ns = C.c_uint32(), C.c_int(), (C.c_double * 3)()
        check(_lib.lib().kt_oligo_launch_info(self._h, C.byref(w), C.byref(d), ns))
        return {"wgs_per_slot": w.value, "measured": bool(d.value), "ns_per_read": {32: ns[0], 96: ns[1], 200: ns[2]}}

    def selftest_quotient(self, d_lo, d_hi):
        """-> (pairs checked, mismatches, checksum of the IEEE quotients' bits) for all 0 <= c <= d, d_lo <= d <= d_hi"""
        n, bad, chk = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(_lib.lib().kt_selftest_quotient(self._h, int(d_lo), int(d_hi), C.byref(n), C.byref(bad), C.byref(chk)))
        return n.value, bad.value, chk.value

    # -- comp cgr (whole sequence) ---------------------------------------------------------------
    def cgr(self, bases, offsets, n_reads, vecsize, xy, bad_pos=None, mem=KT_MEM_DEVICE):
        """xy: 2 f64 per base; bad_pos: u64 scalar (device pointer in device mode) or None"""
        check(_lib.lib().kt_cgr_points(self._h, _ptr(bases), _ptr(offsets), n_reads, float(vecsize), _ptr(xy),
                                       _ptr(bad_pos), mem))
        return xy

    def cgr_host(self, bases, offsets, vecsize=1):
        """-> (total_bases, 2) f64; raises KmertoolsError(KT_ERR_BADNT) on a byte outside ACGTUacgtu"""
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        total = int(offsets[-1])
        xy = np.zeros((max(total, 1), 2), np.float64)
        self.cgr(bases if bases.size else np.zeros(1, np.uint8), offsets, len(offsets) - 1, vecsize, xy, None,
                 KT_MEM_HOST)
        return xy[:total]

    # -- min: window minimisers ---------------------------------------------------------------------
    def minimisers(self, bases, offsets, n_reads, wsize, msize, ev_offsets, kmers, starts, ends, capacity,
                   mem=KT_MEM_DEVICE):
        """-> number of (minimiser, start, end) triples; capacity 0 only counts (synchronises either way)"""
        n = C.c_uint64()
        check(_lib.lib().kt_minimisers(self._h, _ptr(bases), _ptr(offsets), n_reads, int(wsize), int(msize),
                                       _ptr(ev_offsets), _ptr(kmers), _ptr(starts), _ptr(ends), int(capacity),
                                       C.byref(n), mem))
        return n.value

    def minimisers_host(self, bases, offsets, wsize, msize):
        """-> (ev_offsets u64[n+1], kmers, starts, ends): read i owns [ev_offsets[i], ev_offsets[i+1])"""
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = len(offsets) - 1
        evo = np.zeros(n + 1, np.uint64)
        bb = bases if bases.size else np.zeros(1, np.uint8)
        empty = np.zeros(1, np.uint64)
        cnt = self.minimisers(bb, offsets, n, wsize, msize, evo, empty, empty, empty, 0, KT_MEM_HOST) if n else 0
        k = np.zeros(max(cnt, 1), np.uint64)
        s = np.zeros(max(cnt, 1), np.uint64)
        e = np.zeros(max(cnt, 1), np.uint64)
        if cnt:
            self.minimisers(bb, offsets, n, wsize, msize, evo, k, s, e, cnt, KT_MEM_HOST)
        else:
            evo[:] = 0
        return evo, k[:cnt], s[:cnt], e[:cnt]

    # -- KmerGenerator surface -----------------------------------------------------------
    def kmers(self, bases, offsets, n_reads, k, fwd, rev, valid, mem=KT_MEM_DEVICE):
        check(_lib.lib().kt_kmers(self._h, _ptr(bases), _ptr(offsets), n_reads, k, _ptr(fwd), _ptr(rev),
                                  _ptr(valid), mem))

    def kmers_host(self, bases, offsets, k):
        """-> (fwd, rev, end_index) of every k-mer in positional order"""
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        total = int(offsets[-1])
        fwd = np.zeros(max(total, 1), np.uint64)
        rev = np.zeros(max(total, 1), np.uint64)
        valid = np.zeros(max(total, 1), np.uint8)
        if total:
            self.kmers(bases, offsets, len(offsets) - 1, k, fwd, rev, valid, KT_MEM_HOST)
        idx = np.nonzero(valid[:total])[0]
        return fwd[idx], rev[idx], idx.astype(np.uint64)

    # -- ctr routing -------------------------------------------------------------------------
    def route(self, bases, offsets, n_reads, k, n_owners, keys_out, owner_counts, mem=KT_MEM_DEVICE):
        check(_lib.lib().kt_ctr_route(self._h, _ptr(bases), _ptr(offsets), n_reads, k, n_owners, _ptr(keys_out),
                                      _ptr(owner_counts), mem))

    def route_host(self, bases, offsets, k, n_owners):
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        total = int(offsets[-1])
        keys = np.zeros(max(total, 1), np.uint64)
        counts = np.zeros(n_owners, np.uint64)
        self.route(bases if bases.size else np.zeros(1, np.uint8), offsets, len(offsets) - 1, k, n_owners, keys,
                   counts, KT_MEM_HOST)
        return keys[: int(counts.sum())], counts

    # -- synthetic reads (device only) ----------------------------------------------------------
    def synth_reads(self, seed, n_reads, read_len, bases_dev, offsets_dev=None, noise=False, genome_len=0,
                    first_read=0):
        check(_lib.lib().kt_synth_reads(self._h, seed, first_read, n_reads, read_len, int(bool(noise)),
                                        genome_len, _ptr(bases_dev), _ptr(offsets_dev)))


class Counter:
    """HBM-resident canonical k-mer table (kt_ctr) = the reference's CountComputer state."""

    def __init__(self, ctx, k, capacity_slots):
        self.ctx = ctx
        self.k = k
        self._h = C.c_void_p()
        check(_lib.lib().kt_ctr_create(ctx._h, k, int(capacity_slots), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().kt_ctr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        check(_lib.lib().kt_ctr_clear(self._h))

    def add_reads(self, bases, offsets, n_reads, mem=KT_MEM_DEVICE):
        check(_lib.lib().kt_ctr_add_reads(self._h, _ptr(bases), _ptr(offsets), n_reads, mem))

    def add_reads_host(self, bases, offsets, n_parts=1, part=0):
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        self.add_reads_part(bases if bases.size else np.zeros(1, np.uint8), offsets, len(offsets) - 1, n_parts, part,
                            KT_MEM_HOST)

    def add_reads_part(self, bases, offsets, n_reads, n_parts, part, mem=KT_MEM_DEVICE):
        """only the k-mers of hash partition `part` of `n_parts` (out-of-core passes)"""
        check(_lib.lib().kt_ctr_add_reads_part(self._h, _ptr(bases), _ptr(offsets), n_reads, mem, n_parts, part))

    def add_pairs(self, keys, counts, n, mem=KT_MEM_DEVICE):
        check(_lib.lib().kt_ctr_add_pairs(self._h, _ptr(keys), _ptr(counts), n, mem))

    def add_pairs_host(self, keys, counts=None):
        keys = np.ascontiguousarray(keys, np.uint64)
        if counts is not None:
            counts = np.ascontiguousarray(counts, np.uint32)
        if len(keys):
            self.add_pairs(keys, counts, len(keys), KT_MEM_HOST)

    def size(self):
        n = C.c_uint64()
        check(_lib.lib().kt_ctr_size(self._h, C.byref(n)))
        return n.value

    def capacity(self):
        """slots the table really has (the request rounded up to 5..8 eighths of a power of two)"""
        n = C.c_uint64()
        check(_lib.lib().kt_ctr_capacity(self._h, C.byref(n)))
        return n.value

    def export(self, keys, counts, max_out, mem=KT_MEM_DEVICE):
        n = C.c_uint64()
        check(_lib.lib().kt_ctr_export(self._h, _ptr(keys), _ptr(counts), max_out, C.byref(n), mem))
        return n.value

    def export_target(self, keys, counts, max_out):
        """device arrays that the next whole-batch count into the empty table writes its (key, count) pairs to
        (sticky; None, None switches it off): export(keys, counts, ...) afterwards copies nothing.  The library keeps the
        raw pointers until the target is switched off or the counter closed, so the tensors are kept alive here."""
        check(_lib.lib().kt_ctr_export_target(self._h, _ptr(keys), _ptr(counts), int(max_out) if keys is not None else 0))
        self._xt = (keys, counts) if keys is not None else None

    # -- cov: per-read coverage histograms against this table ---------------------------------
    def cov(self, bases, offsets, n_reads, bin_size, bin_count, out, norm=True, dtype="f64", mem=KT_MEM_DEVICE):
        check(_lib.lib().kt_cov_batch(self._h, _ptr(bases), _ptr(offsets), n_reads, int(bin_size), int(bin_count),
                                      int(bool(norm)), _DT[dtype], _ptr(out), mem))
        return out

    def lookup(self, keys, n, counts, mem=KT_MEM_DEVICE):
        """counts[i] = occurrences of the canonical k-mer keys[i] (0: absent) - u64 keys, u32 counts"""
        check(_lib.lib().kt_ctr_lookup(self._h, _ptr(keys), int(n), _ptr(counts), mem))
        return counts

    def lookup_host(self, keys):
        keys = np.ascontiguousarray(keys, np.uint64)
        out = np.zeros(len(keys), np.uint32)
        if len(keys):
            self.lookup(keys, len(keys), out, KT_MEM_HOST)
        return out

    def cov_part(self, bases, offsets, n_reads, bin_size, bin_count, counts, n_parts=1, part=0, mem=KT_MEM_DEVICE):
        """adds the u32 bin counts of the k-mers this table answers for (hash partition `part` of n_parts; a shard: the
        k-mers it owns) to `counts` (n_reads x bin_count)"""
        check(_lib.lib().kt_cov_batch_part(self._h, _ptr(bases), _ptr(offsets), n_reads, int(bin_size), int(bin_count),
                                           _ptr(counts), mem, int(n_parts), int(part)))
        return counts

    def cov_host(self, bases, offsets, bin_size, bin_count, norm=True, dtype="f64"):
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = len(offsets) - 1
        out = np.zeros((n, int(bin_count)), _NP[_DT[dtype]])
        if n:
            self.cov(bases if bases.size else np.zeros(1, np.uint8), offsets, n, bin_size, bin_count, out, norm,
                     dtype, KT_MEM_HOST)
        return out

    def export_host(self, sort=True):
        n = self.size()
        keys = np.zeros(max(n, 1), np.uint64)
        counts = np.zeros(max(n, 1), np.uint32)
        got = self.export(keys, counts, n, KT_MEM_HOST) if n else 0
        keys, counts = keys[:got], counts[:got]
        if sort:
            order = np.argsort(keys, kind="stable")
            keys, counts = keys[order], counts[order]
        return keys, counts


class Sharded:
    """This rank's part of a table sharded over n_ranks GPUs by hash prefix (kt_sharded).  add_reads and finalize are
    collective: every rank calls them the same number of times.  `transport`: ("rccl", id128 bytes) or
    ("host", alltoall) where alltoall(send_addr, recv_addr, bytes_per_rank) moves host memory and returns 0.
    connect=False: allocate only (kt_sharded_create_local); the caller agrees with its peers and calls connect()."""

    def __init__(self, ctx, k, capacity_slots, max_batch_bases, n_ranks=1, rank=0, transport=None, connect=True):
        self.ctx, self.k, self.n_ranks, self.rank = ctx, k, n_ranks, rank
        self._h = C.c_void_p()
        self._cb = None
        self._cb_exc = None
        self._transport = transport
        L = _lib.lib()
        if n_ranks > 1 and transport is None:
            raise ValueError("a sharded counter over several ranks needs a transport")
        if transport is not None and transport[0] not in ("rccl", "host"):
            raise ValueError("unknown transport %r" % (transport[0],))
        check(L.kt_sharded_create_local(ctx._h, k, int(capacity_slots), int(max_batch_bases), n_ranks, rank,
                                        C.byref(self._h)))
        t = C.c_void_p()
        check(L.kt_sharded_table(self._h, C.byref(t)))
        self.table = Counter.__new__(Counter)   # a view of the shard: owned by the kt_sharded
        self.table.ctx, self.table.k, self.table._h = ctx, k, t
        self.table.close = lambda: None
        if connect:
            self.connect()

    def connect(self):
        """brings the transport up (RCCL: ncclCommInitRank - every rank must get here)"""
        L = _lib.lib()
        transport = self._transport
        if self.n_ranks == 1 or transport[0] == "rccl":
            idb = None if self.n_ranks == 1 else (C.c_uint8 * 128).from_buffer_copy(bytes(transport[1]))
            check(L.kt_sharded_connect_rccl(self._h, idb))
        else:
            fn = transport[1]

            def cb(user, send, recv, nbytes):
                # an exception must not be swallowed by ctypes (it would return 0 = success and the library would count
                # whatever the receive buffer held): report failure, keep the exception for the caller
                try:
                    return int(fn(send, recv, nbytes))
                except BaseException as e:  # noqa: BLE001
                    self._cb_exc = e
                    return 1
            self._cb = _lib.ALLTOALL_FN(cb)
            check(L.kt_sharded_connect_host(self._h, C.cast(self._cb, C.c_void_p), None))

    def _checked(self, rc):
        if self._cb_exc is not None:
            e, self._cb_exc = self._cb_exc, None
            raise e
        check(rc)

    def owner_of(self, kmer):
        o = C.c_uint32()
        check(_lib.lib().kt_sharded_owner_of(self._h, int(kmer), C.byref(o)))
        return o.value

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        check(_lib.lib().kt_rccl_unique_id(buf))
        return bytes(buf)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.table._h = C.c_void_p()
            _lib.lib().kt_sharded_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        check(_lib.lib().kt_sharded_clear(self._h))

    def add_reads(self, bases, offsets, n_reads, mem=KT_MEM_DEVICE):
        self._checked(_lib.lib().kt_sharded_add_reads(self._h, _ptr(bases), _ptr(offsets), n_reads, mem))

    def add_reads_host(self, bases, offsets):
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        self.add_reads(bases if bases.size else np.zeros(1, np.uint8), offsets, len(offsets) - 1, KT_MEM_HOST)

    def finalize(self):
        self._checked(_lib.lib().kt_sharded_finalize(self._h))

    def exchanged_bytes(self):
        n = C.c_uint64()
        check(_lib.lib().kt_sharded_exchanged_bytes(self._h, C.byref(n)))
        return n.value

    def route_stats(self):
        """-> (records, kmers): numpy arrays per owner, what the last batch's route pass put into every owner's region"""
        n = C.c_uint32()
        rec = np.zeros(64, np.uint64)
        km = np.zeros(64, np.uint64)
        check(_lib.lib().kt_sharded_route_stats(self._h, C.byref(n), rec.ctypes.data_as(C.c_void_p), km.ctypes.data_as(C.c_void_p)))
        return rec[:n.value], km[:n.value]

    def comm_info(self):
        """-> dict(n_ranks, rccl_ranks, transport): rccl_ranks is ncclCommCount of the library's own communicator
        (0 without one), transport "none" / "rccl" / "host" (kt_sharded_comm_info)"""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(_lib.lib().kt_sharded_comm_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"n_ranks": a.value, "rccl_ranks": b.value, "transport": ("none", "rccl", "host")[c.value]}
