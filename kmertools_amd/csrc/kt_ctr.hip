// kt_ctr.hip - canonical k-mer counting in an HBM-resident table, plus the stand-alone hash
// partition of a batch's k-mers (kt_ctr_route: what the out-of-core passes split by), plus the
// debug k-mer generator surface (kt_kmers).
//
// Replaces CountComputer::count_chunk / merge (reference counter/src/lib.rs:92-234):
// the reference's n_parts concurrent hash maps, chunked spill to text files and
// per-partition re-merge collapse into one open-addressing table that stays in HBM
// (16-byte slots {u64 key, u32 count, pad}: the 64-bit CAS that claims a slot and the 32-bit
// atomic add that counts it touch the same 64-byte line, which halves the DRAM lines moved
// per insert compared with separate key/count arrays - profiles/r1_ctr_*).  HBM-bound
// random access; no MFMA.
#include <vector>

#include "kt_internal.hpp"
#include "kt_launch.hpp"
#include "kt_segment.hpp"
#include "kt_table.hpp"

namespace {

using ktseg::SegArgs;
using ktseg::SegShared;

constexpr int BLOCK = ktseg::BLOCK;

// ---- table primitives: kt_table.hpp -------------------------------------------------

using kttab::Slot;
using kttab::TableRef;
using kttab::table_add;

// new keys claimed by the thread -> the table's distinct counter: one atomic per wave, at the end of the kernel
__device__ __forceinline__ void publish_fresh(uint32_t fresh, uint64_t *distinct) {
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_down(fresh, o, 64);
    if ((threadIdx.x & 63) == 0 && fresh)
        atomicAdd(reinterpret_cast<unsigned long long *>(distinct), (unsigned long long)fresh);
}

__global__ __launch_bounds__(BLOCK) void count_reads_kernel(SegArgs a, TableRef t, uint32_t n_parts, uint32_t part,
                                                            uint64_t *__restrict__ distinct) {
    __shared__ SegShared sm;
    uint32_t fresh = 0;
    for (uint64_t g = blockIdx.x; g < a.n_seg; g += gridDim.x) {
        ktseg::for_each_kmer(a, g, sm, [&](uint64_t f, uint64_t r, uint64_t) {
            const uint64_t m = f < r ? f : r;  // counter/src/lib.rs:124
            if (n_parts > 1 && ktd::owner_of(m, n_parts) != part) return;  // another pass counts this one
            const uint32_t st = table_add(t, m, 1u);
            if (st == 0u) atomicOr(t.flags, 1u);
            fresh += st == 2u;
        });
    }
    publish_fresh(fresh, distinct);
}

__global__ __launch_bounds__(BLOCK) void add_pairs_kernel(const uint64_t *__restrict__ keys,
                                                          const uint32_t *__restrict__ counts, uint64_t n,
                                                          TableRef t, uint64_t *__restrict__ distinct) {
    uint32_t fresh = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t key = keys[i];
        const uint32_t c = counts ? counts[i] : 1u;
        if (key == KT_EMPTY_KEY) continue;
        const uint32_t st = table_add(t, key, c);
        if (st == 0u) atomicOr(t.flags, 1u);
        fresh += st == 2u;
    }
    publish_fresh(fresh, distinct);
}

__global__ __launch_bounds__(BLOCK) void table_clear_kernel(Slot *__restrict__ slots, uint64_t cap) {
    const uint4 empty = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * BLOCK)
        reinterpret_cast<uint4 *>(slots)[i] = empty;
}

// stream the table, compact occupied slots.  A workgroup takes tiles of 4096 slots (16 coalesced 16-byte loads per
// thread, all in flight at once), ballots every load, scans the 64 (wave, load) popcounts and reserves the tile's
// output range with ONE cursor atomic; every store instruction then writes one contiguous run.  (The first version
// took one atomic per wave and 64 slots: 100 M atomics on one address for a 6.4 G-slot table = 1.2 s.)
#ifndef KT_EXPORT_NT
#define KT_EXPORT_NT 0  // bit 0: non-temporal loads of the slots, bit 1: non-temporal stores of the pairs (measured, see DESIGN)
#endif
#ifndef KT_EXPORT_XPT
#define KT_EXPORT_XPT 16
#endif
constexpr uint32_t XPT = KT_EXPORT_XPT, XTILE = BLOCK * XPT;
__global__ __launch_bounds__(BLOCK) void table_export_kernel(const Slot *__restrict__ slots, uint64_t cap,
                                                             uint64_t *__restrict__ out_keys,
                                                             uint32_t *__restrict__ out_counts, uint64_t max_out,
                                                             uint64_t *__restrict__ cursor) {
    __shared__ uint32_t runs[BLOCK / 64 * XPT];  // occupied slots per (wave, load), then their exclusive prefix
    static_assert(BLOCK / 64 * XPT == 64, "one wave scans the runs, one per lane");
    __shared__ uint64_t tile_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t n_tiles = (cap + XTILE - 1) / XTILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        uint4 v[XPT];
#pragma unroll
        for (uint32_t j = 0; j < XPT; j++) {
            const uint64_t i = tile * XTILE + (uint64_t)j * BLOCK + tid;
#if KT_EXPORT_NT & 1
            typedef uint32_t raw4 __attribute__((ext_vector_type(4)));
            if (i < cap) {
                const raw4 r = __builtin_nontemporal_load(reinterpret_cast<const raw4 *>(slots) + i);
                v[j] = make_uint4(r.x, r.y, r.z, r.w);
            } else {
                v[j] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
            }
#else
            v[j] = i < cap ? reinterpret_cast<const uint4 *>(slots)[i] : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
#endif
        }
        uint64_t bal[XPT];
#pragma unroll
        for (uint32_t j = 0; j < XPT; j++) {
            bal[j] = __ballot((v[j].x & v[j].y) != 0xFFFFFFFFu);  // key != KT_EMPTY_KEY
            if (lane == 0) runs[wave * XPT + j] = (uint32_t)__popcll(bal[j]);
        }
        ktd::lds_barrier();
        if (wave == 0) {  // 64 runs: one wave scans them and reserves the tile's output range
            const uint32_t c = runs[lane];
            uint32_t inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t u = __shfl_up(inc, off, 64);
                if (lane >= (uint32_t)off) inc += u;
            }
            runs[lane] = inc - c;
            if (lane == 63)
                tile_base = inc ? atomicAdd(reinterpret_cast<unsigned long long *>(cursor), (unsigned long long)inc) : 0;
        }
        ktd::lds_barrier();
        const uint64_t base = tile_base;
#pragma unroll
        for (uint32_t j = 0; j < XPT; j++) {
            if ((bal[j] >> lane) & 1ull) {
                const uint64_t pos = base + runs[wave * XPT + j] + __popcll(bal[j] & ((1ull << lane) - 1ull));
                if (pos < max_out) {
#if KT_EXPORT_NT & 2
                    __builtin_nontemporal_store(((uint64_t)v[j].y << 32) | v[j].x, out_keys + pos);
                    __builtin_nontemporal_store(v[j].z + 1u, out_counts + pos);
#else
                    out_keys[pos] = ((uint64_t)v[j].y << 32) | v[j].x;
                    out_counts[pos] = v[j].z + 1u;  // stored value is occurrences - 1
#endif
                }
            }
        }
        ktd::lds_barrier();  // runs[] / tile_base are rewritten by the next tile
    }
}

// ---- routing (multi-GPU ownership) ------------------------------------------------------

constexpr int MAX_OWNERS = 64;

__global__ __launch_bounds__(BLOCK) void route_count_kernel(SegArgs a, uint32_t n_owners,
                                                            uint64_t *__restrict__ owner_counts) {
    __shared__ SegShared sm;
    __shared__ uint32_t cnt[MAX_OWNERS];
    if (threadIdx.x < MAX_OWNERS) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t g = blockIdx.x; g < a.n_seg; g += gridDim.x) {
        ktseg::for_each_kmer(a, g, sm, [&](uint64_t f, uint64_t r, uint64_t) {
            const uint64_t m = f < r ? f : r;
            atomicAdd(&cnt[ktd::owner_of(m, n_owners)], 1u);
        });
        if (threadIdx.x < n_owners && cnt[threadIdx.x]) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&owner_counts[threadIdx.x]),
                      (unsigned long long)cnt[threadIdx.x]);
            cnt[threadIdx.x] = 0;
        }
        __syncthreads();
    }
}

// cursors[o] starts at the exclusive prefix of owner_counts; each workgroup reserves a
// contiguous range per owner for its segment, then its lanes fill the range.
__global__ __launch_bounds__(BLOCK) void route_scatter_kernel(SegArgs a, uint32_t n_owners,
                                                              uint64_t *__restrict__ cursors,
                                                              uint64_t *__restrict__ keys_out) {
    __shared__ SegShared sm;
    __shared__ uint32_t cnt[MAX_OWNERS];
    __shared__ uint64_t base[MAX_OWNERS];
    if (threadIdx.x < MAX_OWNERS) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t g = blockIdx.x; g < a.n_seg; g += gridDim.x) {
        ktseg::for_each_kmer(a, g, sm, [&](uint64_t f, uint64_t r, uint64_t) {
            const uint64_t m = f < r ? f : r;
            atomicAdd(&cnt[ktd::owner_of(m, n_owners)], 1u);
        });
        if (threadIdx.x < n_owners) {
            const uint32_t c = cnt[threadIdx.x];
            base[threadIdx.x] = c ? atomicAdd(reinterpret_cast<unsigned long long *>(&cursors[threadIdx.x]),
                                              (unsigned long long)c)
                                  : 0;
            cnt[threadIdx.x] = 0;
        }
        __syncthreads();
        ktseg::for_each_kmer(a, g, sm, [&](uint64_t f, uint64_t r, uint64_t) {
            const uint64_t m = f < r ? f : r;
            const uint32_t o = ktd::owner_of(m, n_owners);
            const uint32_t slot = atomicAdd(&cnt[o], 1u);
            keys_out[base[o] + slot] = m;
        });
        if (threadIdx.x < MAX_OWNERS) cnt[threadIdx.x] = 0;
        __syncthreads();
    }
}

__global__ void route_prefix_kernel(const uint64_t *__restrict__ owner_counts, uint32_t n_owners,
                                    uint64_t *__restrict__ cursors) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint64_t acc = 0;
        for (uint32_t o = 0; o < n_owners; o++) {
            cursors[o] = acc;
            acc += owner_counts[o];
        }
    }
}

// ---- debug surface -------------------------------------------------------------------------

__global__ __launch_bounds__(BLOCK) void kmers_kernel(SegArgs a, uint64_t *__restrict__ fwd,
                                                      uint64_t *__restrict__ rev, uint8_t *__restrict__ valid) {
    __shared__ SegShared sm;
    for (uint64_t g = blockIdx.x; g < a.n_seg; g += gridDim.x) {
        ktseg::for_each_kmer(a, g, sm, [&](uint64_t f, uint64_t r, uint64_t end) {
            fwd[end] = f;
            rev[end] = r;
            valid[end] = 1;
        });
    }
}

int check_overflow(kt_ctr *ctr) {
    uint32_t flag = 0;
    KT_HIP(hipMemcpyAsync(&flag, ctr->flags, sizeof(uint32_t), hipMemcpyDeviceToHost, ctr->ctx->stream));
    KT_HIP(hipStreamSynchronize(ctr->ctx->stream));
    if (flag & 2u)  // (set by the range build that writes straight into the export target, kt_bulk.hip)
        return kt::fail(KT_ERR_ARG, "kt_ctr_export_target: the arrays are smaller than the table (its contents are lost: "
                                    "kt_ctr_clear, then count again with larger arrays or without a target)");
    if (flag) return kt::fail(KT_ERR_FULL, "k-mer table is full: raise capacity_slots");
    return KT_OK;
}

}  // namespace

namespace ktl {

// ---- host helpers ----------------------------------------------------------------------------

uint32_t grid_for(const kt_ctx *ctx, uint64_t work_items, uint32_t per_cu) {
    uint64_t g = (uint64_t)ctx->n_cu * per_cu;
    if (g > work_items) g = work_items;
    if (g < 1) g = 1;
    return (uint32_t)g;
}

// Builds SegArgs for device-resident CSR input (seg_first lives in ctx scratch s_aux0).
int make_seg_args(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                  uint64_t total_bases, int k, SegArgs *out) {
    const uint64_t n_seg = (total_bases + ktseg::SEG - 1) / ktseg::SEG;
    if (int rc = ctx->s_aux0.reserve((n_seg + 2) * sizeof(uint64_t))) return rc;
    uint64_t *seg_first = (uint64_t *)ctx->s_aux0.p;
    const uint64_t threads = n_reads + 1;
    const uint32_t blocks = (uint32_t)((threads + 255) / 256);
    hipLaunchKernelGGL(ktseg::seg_index_kernel, dim3(blocks), dim3(256), 0, ctx->stream, offsets, n_reads,
                       seg_first, n_seg);
    KT_HIP(hipGetLastError());
    out->bases = bases;
    out->offsets = offsets;
    out->seg_first = seg_first;
    out->n_reads = n_reads;
    out->n_seg = n_seg;
    out->k = (uint32_t)k;
    return KT_OK;
}

// total number of bases = offsets[n_reads]; needs a read-back for device offsets
int total_bases_of(kt_ctx *ctx, const uint64_t *offsets, uint64_t n_reads, int mem, uint64_t *total) {
    if (mem == KT_MEM_HOST) {
        *total = offsets[n_reads];
        return KT_OK;
    }
    KT_HIP(hipMemcpyAsync(total, offsets + n_reads, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    KT_HIP(hipStreamSynchronize(ctx->stream));
    return KT_OK;
}

// copies a host CSR batch into ctx scratch; returns device pointers
int stage_batch(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                const uint8_t **d_bases, const uint64_t **d_offsets) {
    const uint64_t total = offsets[n_reads];
    if (offsets[0] != 0) return kt::fail(KT_ERR_ARG, "offsets[0] must be 0");
    if (int rc = ctx->s_bases.reserve(total + 64)) return rc;
    if (int rc = ctx->s_offsets.reserve((n_reads + 1) * 8)) return rc;
    if (total) KT_HIP(hipMemcpyAsync(ctx->s_bases.p, bases, total, hipMemcpyHostToDevice, ctx->stream));
    KT_HIP(hipMemcpyAsync(ctx->s_offsets.p, offsets, (n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    *d_bases = (const uint8_t *)ctx->s_bases.p;
    *d_offsets = (const uint64_t *)ctx->s_offsets.p;
    return KT_OK;
}

}  // namespace ktl

using namespace ktl;

extern "C" {

}  // extern "C"

extern "C" {

int kt_ctr_create(kt_ctx *ctx, int k, uint64_t capacity_slots, kt_ctr **out) {
    if (!ctx || !out) return kt::fail(KT_ERR_ARG, "kt_ctr_create: null");
    *out = nullptr;
    if (k < 1 || k > 31) return kt::fail(KT_ERR_ARG, "kt_ctr_create: k must be in 1..31");
    if (int rc = ctx->use()) return rc;
    const kttab::Geom geom = kttab::make_geom(capacity_slots, k);
    const uint64_t cap = geom.cap;
    kt_ctr *c = new (std::nothrow) kt_ctr();
    if (!c) return kt::fail(KT_ERR_NOMEM, "kt_ctr_create: host alloc");
    c->ctx = ctx;
    c->k = k;
    c->cap = cap;
    c->shift = geom.shift;
    c->m8 = geom.m8;
    c->kbits = geom.kbits;
    hipError_t e = hipMalloc((void **)&c->slots, cap * sizeof(Slot));
    if (e == hipSuccess) e = hipMalloc((void **)&c->flags, 64);
    if (e == hipSuccess) e = hipMalloc((void **)&c->cursor, 64);
    if (e == hipSuccess) e = hipMalloc((void **)&c->distinct, 64);
    if (e == hipSuccess) e = hipMalloc((void **)&c->range_counts, (cap / geom.range_slots() + 1) * sizeof(uint32_t));
    if (e != hipSuccess) {
        kt_ctr_destroy(c);
        return kt::fail(KT_ERR_NOMEM, std::string("kt_ctr_create: hipMalloc: ") + hipGetErrorString(e));
    }
    *out = c;
    return kt_ctr_clear(c);
}

int kt_ctr_destroy(kt_ctr *ctr) {
    if (!ctr) return KT_OK;
    if (ctr->ctx) {
        (void)hipSetDevice(ctr->ctx->device);
        (void)hipStreamSynchronize(ctr->ctx->stream);
    }
    if (ctr->slots) (void)hipFree(ctr->slots);
    if (ctr->flags) (void)hipFree(ctr->flags);
    if (ctr->cursor) (void)hipFree(ctr->cursor);
    if (ctr->distinct) (void)hipFree(ctr->distinct);
    if (ctr->range_counts) (void)hipFree(ctr->range_counts);
    ctr->b_ext.release();
    ctr->b_stage_k.release();
    ctr->b_stage_c.release();
    ctr->b_keys1.release();
    ctr->b_keys2.release();
    ctr->b_meta.release();
    ctr->b_desc.release();
    ctr->b_pack.release();
    kt_bulk_job_free(ctr->job);
    delete ctr;
    return KT_OK;
}

// Clearing is deferred: the bulk build (kt_bulk.hip) overwrites every slot, so a clear that is
// followed by a whole-batch kt_ctr_add_reads never has to touch the table.
static int ensure_cleared(kt_ctr *ctr) {
    if (ctr->dense) return kt_table_image(ctr);  // (a dense table is never pending a clear)
    if (!ctr->needs_clear) return KT_OK;
    hipLaunchKernelGGL(table_clear_kernel, dim3(grid_for(ctr->ctx, (ctr->cap + BLOCK - 1) / BLOCK, 8)), dim3(BLOCK), 0,
                       ctr->ctx->stream, (Slot *)ctr->slots, ctr->cap);
    KT_HIP(hipGetLastError());
    ctr->needs_clear = false;
    return KT_OK;
}
}  // extern "C"
int ktl::table_ready(kt_ctr *ctr) {
    if (int rc = ensure_cleared(ctr)) return rc;
    return check_overflow(ctr);
}
extern "C" {

int kt_ctr_clear(kt_ctr *ctr) {
    if (!ctr) return kt::fail(KT_ERR_ARG, "kt_ctr_clear: null");
    if (int rc = ctr->ctx->use()) return rc;
    KT_HIP(hipMemsetAsync(ctr->flags, 0, 64, ctr->ctx->stream));
    KT_HIP(hipMemsetAsync(ctr->distinct, 0, 8, ctr->ctx->stream));
    ctr->needs_clear = true;
    ctr->empty = true;
    ctr->dense = false;
    ctr->dense_ext = false;
    ctr->stage_n = 0;  // (what kt_ctr_export_stage staged is gone with the table's contents)
    return KT_OK;
}

int kt_ctr_export_target(kt_ctr *ctr, uint64_t *keys_dev, uint32_t *counts_dev, uint64_t max_out) {
    if (!ctr) return kt::fail(KT_ERR_ARG, "kt_ctr_export_target: null ctr");
    if ((keys_dev == nullptr) != (counts_dev == nullptr))
        return kt::fail(KT_ERR_ARG, "kt_ctr_export_target: both arrays or neither");
    if (int rc = ctr->ctx->use()) return rc;
    // a table whose entries live in the current target gets its own copy (the probing image) before the target moves
    if (ctr->dense_ext && (keys_dev != ctr->xt_keys || counts_dev != ctr->xt_counts))
        if (int rc = kt_table_image(ctr)) return rc;
    ctr->xt_keys = keys_dev;
    ctr->xt_counts = counts_dev;
    ctr->xt_max = keys_dev ? max_out : 0;
    ctr->stage_n = 0;  // (what was staged may have been the old target's arrays)
    return KT_OK;
}

int kt_ctr_add_reads(kt_ctr *ctr, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int mem) {
    return kt_ctr_add_reads_part(ctr, bases, offsets, n_reads, mem, 1, 0);
}

int kt_ctr_add_reads_part(kt_ctr *ctr, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int mem,
                          uint32_t n_parts, uint32_t part) {
    if (!ctr) return kt::fail(KT_ERR_ARG, "kt_ctr_add_reads: null ctr");
    if (n_parts < 1 || part >= n_parts) return kt::fail(KT_ERR_ARG, "kt_ctr_add_reads_part: need part < n_parts");
    if (n_reads == 0) return KT_OK;
    ctr->stage_n = 0;  // the table changes: what kt_ctr_export_stage staged is no longer the table (fetch says so)
    if (!offsets) return kt::fail(KT_ERR_ARG, "kt_ctr_add_reads: null offsets");
    kt_ctx *ctx = ctr->ctx;
    if (int rc = ctx->use()) return rc;
    uint64_t total = 0;
    if (int rc = total_bases_of(ctx, offsets, n_reads, mem, &total)) return rc;
    if (total == 0) return KT_OK;
    if (!bases) return kt::fail(KT_ERR_ARG, "kt_ctr_add_reads: null bases");
    const uint8_t *d_bases = bases;
    const uint64_t *d_offsets = offsets;
    if (mem == KT_MEM_HOST) {
        if (int rc = stage_batch(ctx, bases, offsets, n_reads, &d_bases, &d_offsets)) return rc;
    }
    {
        // a whole batch: partition by hash prefix + one LDS build per range of the table, no global atomics.  Into a
        // table that holds data the ranges are rebuilt from what they have + the batch (worth it for large batches)
        int done = 0;
        if (int rc = kt_bulk_build(ctr, d_bases, d_offsets, n_reads, total, n_parts, part, &done)) return rc;
        if (done) {
            if (mem == KT_MEM_HOST) KT_HIP(hipStreamSynchronize(ctx->stream));
            return KT_OK;
        }
    }
    ctr->empty = false;
    if (int rc = ensure_cleared(ctr)) return rc;
    SegArgs a;
    if (int rc = make_seg_args(ctx, d_bases, d_offsets, n_reads, total, ctr->k, &a)) return rc;
    TableRef t{(Slot *)ctr->slots, ktl::geom_of(ctr), ctr->flags};
    hipLaunchKernelGGL(count_reads_kernel, dim3(grid_for(ctx, a.n_seg, 8)), dim3(BLOCK), 0, ctx->stream, a, t, n_parts,
                       part, ctr->distinct);
    KT_HIP(hipGetLastError());
    if (mem == KT_MEM_HOST) KT_HIP(hipStreamSynchronize(ctx->stream));
    return KT_OK;
}

int kt_ctr_add_pairs(kt_ctr *ctr, const uint64_t *keys, const uint32_t *counts, uint64_t n, int mem) {
    if (!ctr) return kt::fail(KT_ERR_ARG, "kt_ctr_add_pairs: null ctr");
    if (n == 0) return KT_OK;
    if (!keys) return kt::fail(KT_ERR_ARG, "kt_ctr_add_pairs: null keys");
    ctr->stage_n = 0;  // (see kt_ctr_add_reads_part)
    kt_ctx *ctx = ctr->ctx;
    if (int rc = ctx->use()) return rc;
    const uint64_t *d_keys = keys;
    const uint32_t *d_counts = counts;
    if (mem == KT_MEM_HOST) {
        if (int rc = ctx->s_aux1.reserve(n * 8)) return rc;
        KT_HIP(hipMemcpyAsync(ctx->s_aux1.p, keys, n * 8, hipMemcpyHostToDevice, ctx->stream));
        d_keys = (const uint64_t *)ctx->s_aux1.p;
        if (counts) {
            if (int rc = ctx->s_aux2.reserve(n * 4)) return rc;
            KT_HIP(hipMemcpyAsync(ctx->s_aux2.p, counts, n * 4, hipMemcpyHostToDevice, ctx->stream));
            d_counts = (const uint32_t *)ctx->s_aux2.p;
        }
    }
    if (!d_counts) {
        // raw k-mers (routed from other GPUs): the bulk build, no global atomics
        int done = 0;
        if (int rc = kt_bulk_build_keys(ctr, d_keys, n, &done)) return rc;
        if (done) {
            if (mem == KT_MEM_HOST) KT_HIP(hipStreamSynchronize(ctx->stream));
            return KT_OK;
        }
    }
    ctr->empty = false;
    if (int rc = ensure_cleared(ctr)) return rc;
    TableRef t{(Slot *)ctr->slots, ktl::geom_of(ctr), ctr->flags};
    hipLaunchKernelGGL(add_pairs_kernel, dim3(grid_for(ctx, (n + BLOCK - 1) / BLOCK, 8)), dim3(BLOCK), 0,
                       ctx->stream, d_keys, d_counts, n, t, ctr->distinct);
    KT_HIP(hipGetLastError());
    if (mem == KT_MEM_HOST) KT_HIP(hipStreamSynchronize(ctx->stream));
    return KT_OK;
}

}  // extern "C"

// an empty table filled from the (key, occurrences) pairs the export target holds (probing path; the rare road: a
// table that was counted into its export arrays and is then added to or probed after all)
int kt_ctr_reload_pairs(kt_ctr *ctr, const uint64_t *d_keys, const uint32_t *d_counts) {
    kt_ctx *ctx = ctr->ctx;
    if (int rc = ctx->use()) return rc;
    uint64_t n = 0;
    KT_HIP(hipMemcpyAsync(&n, ctr->distinct, 8, hipMemcpyDeviceToHost, ctx->stream));
    KT_HIP(hipStreamSynchronize(ctx->stream));
    if (int rc = check_overflow(ctr)) return rc;  // (arrays that were too small hold nothing usable)
    ctr->needs_clear = true;
    if (int rc = ensure_cleared(ctr)) return rc;
    KT_HIP(hipMemsetAsync(ctr->distinct, 0, 8, ctx->stream));
    if (n == 0) return KT_OK;
    TableRef t{(Slot *)ctr->slots, ktl::geom_of(ctr), ctr->flags};
    hipLaunchKernelGGL(add_pairs_kernel, dim3(grid_for(ctx, (n + BLOCK - 1) / BLOCK, 8)), dim3(BLOCK), 0, ctx->stream,
                       d_keys, d_counts, n, t, ctr->distinct);
    KT_HIP(hipGetLastError());
    return KT_OK;
}

extern "C" {

int kt_ctr_capacity(kt_ctr *ctr, uint64_t *slots) {
    if (!ctr || !slots) return kt::fail(KT_ERR_ARG, "kt_ctr_capacity: null");
    *slots = ctr->cap;
    return KT_OK;
}

int kt_ctr_size(kt_ctr *ctr, uint64_t *distinct) {
    if (!ctr || !distinct) return kt::fail(KT_ERR_ARG, "kt_ctr_size: null");
    kt_ctx *ctx = ctr->ctx;
    if (int rc = ctx->use()) return rc;
    // every insert path keeps the count of occupied slots (claims in the probing kernels, placed keys in the range
    // builds), so the size is one 8-byte read - not a scan of the table
    KT_HIP(hipMemcpyAsync(distinct, ctr->distinct, 8, hipMemcpyDeviceToHost, ctx->stream));
    KT_HIP(hipStreamSynchronize(ctx->stream));
    return check_overflow(ctr);
}

int kt_ctr_export(kt_ctr *ctr, uint64_t *keys, uint32_t *counts, uint64_t max_out, uint64_t *n_out, int mem) {
    if (!ctr || !n_out) return kt::fail(KT_ERR_ARG, "kt_ctr_export: null");
    if (max_out && (!keys || !counts)) return kt::fail(KT_ERR_ARG, "kt_ctr_export: null output");
    kt_ctx *ctx = ctr->ctx;
    if (int rc = ctx->use()) return rc;
    if (!ctr->dense)
        if (int rc = ensure_cleared(ctr)) return rc;
    if (int rc = check_overflow(ctr)) return rc;
    uint64_t *d_keys = keys;
    uint32_t *d_counts = counts;
    if (mem == KT_MEM_HOST && max_out) {
        if (int rc = ctx->s_aux1.reserve(max_out * 8)) return rc;
        if (int rc = ctx->s_aux2.reserve(max_out * 4)) return rc;
        d_keys = (uint64_t *)ctx->s_aux1.p;
        d_counts = (uint32_t *)ctx->s_aux2.p;
    }
    if (ctr->dense) {  // packed ranges: a coalesced copy, no compaction
        uint64_t n = 0;
        if (ctr->dense_ext) {  // ... or no copy at all: the build wrote the entries to the export target
            d_keys = mem == KT_MEM_HOST ? ctr->xt_keys : keys;
            d_counts = mem == KT_MEM_HOST ? ctr->xt_counts : counts;
        }
        if (int rc = kt_table_dense_export(ctr, d_keys, d_counts, max_out, &n)) return rc;
        const uint64_t written = n < max_out ? n : max_out;
        if (mem == KT_MEM_HOST && written) {
            KT_HIP(hipMemcpy(keys, d_keys, written * 8, hipMemcpyDeviceToHost));
            KT_HIP(hipMemcpy(counts, d_counts, written * 4, hipMemcpyDeviceToHost));
        }
        *n_out = written;
        if (n > max_out) return kt::fail(KT_ERR_ARG, "kt_ctr_export: max_out smaller than the table's size");
        return KT_OK;
    }
    KT_HIP(hipMemsetAsync(ctr->cursor, 0, 8, ctx->stream));
    hipLaunchKernelGGL(table_export_kernel, dim3(grid_for(ctx, (ctr->cap + XTILE - 1) / XTILE, 8)), dim3(BLOCK), 0,
                       ctx->stream, (const Slot *)ctr->slots, ctr->cap, d_keys, d_counts, max_out, ctr->cursor);
    KT_HIP(hipGetLastError());
    uint64_t n = 0;
    KT_HIP(hipMemcpyAsync(&n, ctr->cursor, 8, hipMemcpyDeviceToHost, ctx->stream));
    KT_HIP(hipStreamSynchronize(ctx->stream));
    const uint64_t written = n < max_out ? n : max_out;
    if (mem == KT_MEM_HOST && written) {
        KT_HIP(hipMemcpy(keys, d_keys, written * 8, hipMemcpyDeviceToHost));
        KT_HIP(hipMemcpy(counts, d_counts, written * 4, hipMemcpyDeviceToHost));
    }
    *n_out = written;
    if (n > max_out) return kt::fail(KT_ERR_ARG, "kt_ctr_export: max_out smaller than the table's size");
    return KT_OK;
}

// The table's entries in pieces (out-of-core callers: the host never holds more than one piece).  kt_ctr_export_stage
// puts all of them into a device-side staging area of the library (for a table counted into an export target: they are
// there already) and reports how many; kt_ctr_export_fetch copies entries [first, first + count) to host arrays.
int kt_ctr_export_stage(kt_ctr *ctr, uint64_t *n_out) {
    if (!ctr || !n_out) return kt::fail(KT_ERR_ARG, "kt_ctr_export_stage: null");
    kt_ctx *ctx = ctr->ctx;
    if (int rc = ctx->use()) return rc;
    uint64_t n = 0;
    if (int rc = kt_ctr_size(ctr, &n)) return rc;
    ctr->stage_keys = nullptr;
    ctr->stage_counts = nullptr;
    ctr->stage_n = 0;
    if (n) {
        if (ctr->dense && ctr->dense_ext) {
            ctr->stage_keys = ctr->xt_keys;
            ctr->stage_counts = ctr->xt_counts;
        } else {
            // (the table's own staging buffers - the context's scratch is re-reserved by any other call on the context:
            // host-memory adds and exports, routing, cov, minimisers - ADVICE r4)
            if (int rc = ctr->b_stage_k.reserve(n * 8)) return rc;
            if (int rc = ctr->b_stage_c.reserve(n * 4)) return rc;
            uint64_t got = 0;
            if (int rc = kt_ctr_export(ctr, (uint64_t *)ctr->b_stage_k.p, (uint32_t *)ctr->b_stage_c.p, n, &got, KT_MEM_DEVICE)) return rc;
            n = got;
            ctr->stage_keys = (const uint64_t *)ctr->b_stage_k.p;
            ctr->stage_counts = (const uint32_t *)ctr->b_stage_c.p;
        }
    }
    ctr->stage_n = n;
    *n_out = n;
    return KT_OK;
}

int kt_ctr_export_fetch(kt_ctr *ctr, uint64_t first, uint64_t count, uint64_t *keys_host, uint32_t *counts_host) {
    if (!ctr) return kt::fail(KT_ERR_ARG, "kt_ctr_export_fetch: null");
    if (first > ctr->stage_n || count > ctr->stage_n - first)
        return kt::fail(KT_ERR_ARG, "kt_ctr_export_fetch: beyond the staged entries (call kt_ctr_export_stage first)");
    if (!count) return KT_OK;
    if (!keys_host || !counts_host) return kt::fail(KT_ERR_ARG, "kt_ctr_export_fetch: null output");
    kt_ctx *ctx = ctr->ctx;
    if (int rc = ctx->use()) return rc;
    KT_HIP(hipMemcpyAsync(keys_host, ctr->stage_keys + first, count * 8, hipMemcpyDeviceToHost, ctx->stream));
    KT_HIP(hipMemcpyAsync(counts_host, ctr->stage_counts + first, count * 4, hipMemcpyDeviceToHost, ctx->stream));
    KT_HIP(hipStreamSynchronize(ctx->stream));
    return KT_OK;
}

int kt_ctr_route(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int k,
                 int n_owners, uint64_t *keys_out, uint64_t *owner_counts, int mem) {
    if (!ctx || !owner_counts) return kt::fail(KT_ERR_ARG, "kt_ctr_route: null");
    if (k < 1 || k > 31) return kt::fail(KT_ERR_ARG, "kt_ctr_route: k must be in 1..31");
    if (n_owners < 1 || n_owners > MAX_OWNERS) return kt::fail(KT_ERR_ARG, "kt_ctr_route: n_owners must be in 1..64");
    if (int rc = ctx->use()) return rc;
    // device scalars: [0..63] owner counts, [64..127] cursors
    if (int rc = ctx->s_aux2.reserve(2 * MAX_OWNERS * 8)) return rc;
    uint64_t *d_counts = (uint64_t *)ctx->s_aux2.p;
    uint64_t *d_cursors = d_counts + MAX_OWNERS;
    KT_HIP(hipMemsetAsync(d_counts, 0, 2 * MAX_OWNERS * 8, ctx->stream));
    uint64_t total = 0;
    if (n_reads) {
        if (!offsets) return kt::fail(KT_ERR_ARG, "kt_ctr_route: null offsets");
        if (int rc = total_bases_of(ctx, offsets, n_reads, mem, &total)) return rc;
    }
    uint64_t *d_keys = keys_out;
    if (total) {
        if (!bases || !keys_out) return kt::fail(KT_ERR_ARG, "kt_ctr_route: null buffer");
        const uint8_t *d_bases = bases;
        const uint64_t *d_offsets = offsets;
        if (mem == KT_MEM_HOST) {
            if (int rc = stage_batch(ctx, bases, offsets, n_reads, &d_bases, &d_offsets)) return rc;
            if (int rc = ctx->s_aux1.reserve(total * 8)) return rc;
            d_keys = (uint64_t *)ctx->s_aux1.p;
        }
        SegArgs a;
        if (int rc = make_seg_args(ctx, d_bases, d_offsets, n_reads, total, k, &a)) return rc;
        const uint32_t grid = grid_for(ctx, a.n_seg, 8);
        hipLaunchKernelGGL(route_count_kernel, dim3(grid), dim3(BLOCK), 0, ctx->stream, a, (uint32_t)n_owners, d_counts);
        hipLaunchKernelGGL(route_prefix_kernel, dim3(1), dim3(64), 0, ctx->stream, d_counts, (uint32_t)n_owners, d_cursors);
        hipLaunchKernelGGL(route_scatter_kernel, dim3(grid), dim3(BLOCK), 0, ctx->stream, a, (uint32_t)n_owners,
                           d_cursors, d_keys);
        KT_HIP(hipGetLastError());
    }
    if (mem == KT_MEM_DEVICE) {
        KT_HIP(hipMemcpyAsync(owner_counts, d_counts, n_owners * 8, hipMemcpyDeviceToDevice, ctx->stream));
        return KT_OK;
    }
    KT_HIP(hipMemcpyAsync(owner_counts, d_counts, n_owners * 8, hipMemcpyDeviceToHost, ctx->stream));
    KT_HIP(hipStreamSynchronize(ctx->stream));
    uint64_t n_keys = 0;
    for (int o = 0; o < n_owners; o++) n_keys += owner_counts[o];
    if (n_keys) KT_HIP(hipMemcpy(keys_out, d_keys, n_keys * 8, hipMemcpyDeviceToHost));
    return KT_OK;
}

int kt_kmers(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int k, uint64_t *fwd,
             uint64_t *rev, uint8_t *valid, int mem) {
    if (!ctx) return kt::fail(KT_ERR_ARG, "kt_kmers: null ctx");
    if (k < 1 || k > 31) return kt::fail(KT_ERR_ARG, "kt_kmers: k must be in 1..31");
    if (n_reads == 0) return KT_OK;
    if (!offsets) return kt::fail(KT_ERR_ARG, "kt_kmers: null offsets");
    if (int rc = ctx->use()) return rc;
    uint64_t total = 0;
    if (int rc = total_bases_of(ctx, offsets, n_reads, mem, &total)) return rc;
    if (total == 0) return KT_OK;
    if (!bases || !fwd || !rev || !valid) return kt::fail(KT_ERR_ARG, "kt_kmers: null buffer");
    const uint8_t *d_bases = bases;
    const uint64_t *d_offsets = offsets;
    uint64_t *d_fwd = fwd, *d_rev = rev;
    uint8_t *d_valid = valid;
    if (mem == KT_MEM_HOST) {
        if (int rc = stage_batch(ctx, bases, offsets, n_reads, &d_bases, &d_offsets)) return rc;
        if (int rc = ctx->s_out.reserve(total * 17)) return rc;
        d_fwd = (uint64_t *)ctx->s_out.p;
        d_rev = d_fwd + total;
        d_valid = (uint8_t *)(d_rev + total);
    }
    KT_HIP(hipMemsetAsync(d_valid, 0, total, ctx->stream));
    SegArgs a;
    if (int rc = make_seg_args(ctx, d_bases, d_offsets, n_reads, total, k, &a)) return rc;
    hipLaunchKernelGGL(kmers_kernel, dim3(grid_for(ctx, a.n_seg, 8)), dim3(BLOCK), 0, ctx->stream, a, d_fwd, d_rev,
                       d_valid);
    KT_HIP(hipGetLastError());
    if (mem == KT_MEM_HOST) {
        KT_HIP(hipMemcpyAsync(fwd, d_fwd, total * 8, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipMemcpyAsync(rev, d_rev, total * 8, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipMemcpyAsync(valid, d_valid, total, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipStreamSynchronize(ctx->stream));
    }
    return KT_OK;
}

}  // extern "C"
