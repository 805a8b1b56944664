// kt_cov.hip - per-read k-mer coverage histograms against the HBM-resident count table.
//
// Replaces CovComputer::vectorise_one (reference coverage/src/lib.rs:165-184): for every
// canonical k-mer of a read, look its count up (0 when absent), bin = min(count / bin_size,
// bin_count - 1), histogram the bins, optionally divide by max(1, #k-mers).  The reference
// re-reads kmers.counts into a host HashMap; here the table built by kt_ctr_add_reads is
// probed where it lies.  HBM-bound random 16-byte reads; no MFMA.
//
// Same segment front-end as the counting kernels.  A thread walks its 32 window starts in groups:
// generate GROUP canonical k-mers, issue their home-slot loads together, then add into a per-segment LDS image
// of the bin rows of the reads that touch the segment (row = read id - first read of the
// segment); the image is flushed with one global atomic per non-zero cell, so a read that
// straddles segments is still summed exactly.  Segments made of very many tiny reads (image
// larger than the LDS budget) add to global memory directly.
#include "kt_internal.hpp"
#include "kt_launch.hpp"
#include "kt_segment.hpp"
#include "kt_table.hpp"

#ifndef KT_COV_GROUP
#define KT_COV_GROUP 8
#endif

namespace {

using ktseg::SegArgs;
using ktseg::SegShared;
using kttab::Slot;

constexpr int BLOCK = ktseg::BLOCK;
constexpr uint32_t GROUP = KT_COV_GROUP;  // table probes in flight per thread
constexpr uint32_t ROWS_LDS = 6144;  // u32 cells of bin rows staged per segment (24 KB)

struct CovArgs {
    const Slot *slots;
    kttab::Geom g;       // capacity and hash -> home slot mapping
    uint32_t bin_size;   // 0 = wider than any u32 count: every k-mer falls in bin 0
    uint32_t bin_count;
    uint32_t *counts;    // n_reads x bin_count, zeroed
    uint32_t n_parts, part;  // n_parts > 1: only the k-mers of hash partition `part` are binned (kt_cov_batch_part)
    uint32_t shard;          // the table is a shard of a sharded table (1; 2: the first shard - see cov_kernel): the rows are summed over the shards
};

__device__ __forceinline__ uint4 load_slot(const Slot *slots, uint64_t slot) {
    return *reinterpret_cast<const uint4 *>(slots + slot);
}

// occurrences of `key` given the already-loaded home slot `v`; walks on (round the key's range, kt_table.hpp)
// only on a collision - the probe sequence is recomputed then, so that the common case carries no state for it
__device__ __forceinline__ uint32_t resolve_count(const CovArgs &c, uint4 v, uint64_t key) {
    uint64_t kk = ((uint64_t)v.y << 32) | v.x;
    if (kk == key) return v.z + 1u;  // stored value is occurrences - 1
    if (kk == KT_EMPTY_KEY) return 0u;
    kttab::Probe p = kttab::probe_of(key, c.g);
    for (uint32_t probe = 1; probe < p.rs; probe++) {
        p.next();
        v = load_slot(c.slots, p.slot());
        kk = ((uint64_t)v.y << 32) | v.x;
        if (kk == key) return v.z + 1u;
        if (kk == KT_EMPTY_KEY) return 0u;
    }
    return 0u;
}

__global__ __launch_bounds__(BLOCK) void cov_kernel(SegArgs a, CovArgs c) {
    __shared__ SegShared sm;
    __shared__ uint32_t rows[ROWS_LDS];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < ROWS_LDS; i += BLOCK) rows[i] = 0;
    const uint32_t last_bin = c.bin_count - 1u;

    for (uint64_t g = blockIdx.x; g < a.n_seg; g += gridDim.x) {
        ktseg::stage_segment(a, g, sm);  // its barriers also order the rows[] zeroing
        ktseg::Window w(sm, tid, a.k);
        uint32_t ok = 0;
        for (uint32_t j = 0; j < ktseg::PER_THREAD; j++) ok |= (w.ok(j) ? 1u : 0u) << j;

        // reads that can own a k-mer starting in this segment: [rbase, r_hi)
        const uint64_t B0 = g * ktseg::SEG;
        const uint64_t r_first = a.seg_first[g];
        const uint64_t r_hi = a.seg_first[g + 1];
        const uint64_t rbase = r_first ? r_first - 1 : 0;  // offsets[rbase] <= B0
        const uint64_t cells = (r_hi - rbase) * (uint64_t)c.bin_count;
        const bool in_lds = cells <= ROWS_LDS;

        if (ok) {
            // read id of the thread's first valid window start, then walk forward
            const uint64_t s0 = B0 + (uint64_t)ktseg::PER_THREAD * tid;
            uint64_t lo = rbase, hi = r_hi;
            {
                const uint64_t s = s0 + (uint32_t)__builtin_ctz(ok);
                while (hi - lo > 1) {
                    const uint64_t mid = (lo + hi) >> 1;
                    if (a.offsets[mid] <= s) lo = mid; else hi = mid;
                }
            }
            uint64_t rid = lo;
            uint64_t next = a.offsets[rid + 1];
            // GROUP probes in flight per thread; rolled so the k-mers never sit in registers all at once
#pragma unroll 1
            for (uint32_t jj = 0; jj < ktseg::PER_THREAD; jj += GROUP) {
                uint64_t key[GROUP];
                uint4 v[GROUP];
#pragma unroll
                for (uint32_t u = 0; u < GROUP; u++) {
                    key[u] = w.f < w.r ? w.f : w.r;
                    w.step();
                    v[u] = load_slot(c.slots, kttab::probe_of(key[u], c.g).slot());
                }
#pragma unroll
                for (uint32_t u = 0; u < GROUP; u++) {
                    if (!((ok >> (jj + u)) & 1u)) continue;
                    // (a k-mer that another pass / another shard answers for is not "absent" here: it is skipped)
                    if (c.n_parts > 1 && ktd::owner_of(key[u], c.n_parts) != c.part) continue;
                    const uint32_t cnt = resolve_count(c, v[u], key[u]);
                    uint32_t bin = c.bin_size ? cnt / c.bin_size : 0u;  // coverage/src/lib.rs:172
                    bin = bin < last_bin ? bin : last_bin;              // :173
                    const uint64_t s = s0 + jj + u;
                    while (s >= next) next = a.offsets[++rid + 1];      // empty reads are stepped over
                    auto add = [&](uint32_t b, uint32_t n) {
                        if (in_lds) atomicAdd(&rows[(uint32_t)(rid - rbase) * c.bin_count + b], n);
                        else atomicAdd(&c.counts[rid * c.bin_count + b], n);
                    };
                    if (!c.shard) {
                        add(bin, 1u);
                    } else {
                        // one shard of a table spread over several GPUs by minimiser (kt_shard.hip): a k-mer this shard does
                        // not hold is not "absent" - it may be elsewhere.  Shard 0's pass puts every k-mer into bin 0 (as if
                        // absent everywhere); the one shard that holds a k-mer moves it from there to its bin.  The shards'
                        // rows, summed modulo 2^32 (u32 cells), are the rows of the whole table.
                        if (c.shard == 2u) add(0u, 1u);
                        if (cnt && bin) {
                            add(bin, 1u);
                            add(0u, 0xFFFFFFFFu);
                        }
                    }
                }
            }
        }
        __syncthreads();  // sm is restaged by the next segment; rows[] is complete
        if (in_lds) {
            uint32_t *dst = c.counts + rbase * c.bin_count;
            for (uint32_t i = tid; i < (uint32_t)cells; i += BLOCK) {
                const uint32_t n = rows[i];
                if (n) {
                    atomicAdd(&dst[i], n);
                    rows[i] = 0;
                }
            }
        }
    }
}

// occurrences of keys[i] in the table (0: absent), one thread per key
__global__ __launch_bounds__(BLOCK) void lookup_kernel(const Slot *__restrict__ slots, kttab::Geom g, const uint64_t *__restrict__ keys,
                                                       uint64_t n, uint32_t *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t key = keys[i];
        kttab::Probe p = kttab::probe_of(key, g);
        uint32_t cnt = 0;
        for (uint32_t probe = 0; probe < p.rs; probe++) {
            const uint4 v = load_slot(slots, p.slot());
            const uint64_t kk = ((uint64_t)v.y << 32) | v.x;
            if (kk == key) {
                cnt = v.z + 1u;
                break;
            }
            if (kk == KT_EMPTY_KEY) break;
            p.next();
        }
        out[i] = key == KT_EMPTY_KEY ? 0u : cnt;
    }
}

// one thread per read: total = sum of the row, out = count / max(1, total) (:180-182)
template <class T>
__global__ __launch_bounds__(BLOCK) void cov_finalize_kernel(const uint32_t *__restrict__ counts, uint64_t n_reads,
                                                             uint32_t bin_count, int norm, T *__restrict__ out) {
    const uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t *row = counts + r * bin_count;
    T *o = out + r * bin_count;
    double d = 1.0;
    if (norm) {
        uint64_t total = 0;
        for (uint32_t b = 0; b < bin_count; b++) total += row[b];
        d = total > 1 ? (double)total : 1.0;
    }
    for (uint32_t b = 0; b < bin_count; b++) {
        const double x = (double)row[b];
        o[b] = (T)(norm ? x / d : x);
    }
}

}  // namespace

using namespace ktl;

// the lookup pass: u32 bin counts of the reads' k-mers (those of hash partition `part` of n_parts) into d_counts
static int cov_counts(kt_ctr *table, kt_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                      uint64_t total, uint64_t bin_size, uint64_t bin_count, uint32_t *d_counts, uint32_t n_parts,
                      uint32_t part) {
    if (!total) return KT_OK;
    SegArgs a;
    if (int rc = make_seg_args(ctx, d_bases, d_offsets, n_reads, total, table->k, &a)) return rc;
    CovArgs c{(const Slot *)table->slots, ktl::geom_of(table), bin_size > 0xFFFFFFFFull ? 0u : (uint32_t)bin_size,
              (uint32_t)bin_count, d_counts, n_parts, part, table->n_owners > 1 ? (table->owner == 0 ? 2u : 1u) : 0u};
    hipLaunchKernelGGL(cov_kernel, dim3(grid_for(ctx, a.n_seg, 8)), dim3(BLOCK), 0, ctx->stream, a, c);
    KT_HIP(hipGetLastError());
    return KT_OK;
}

extern "C" int kt_cov_batch_part(kt_ctr *table, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                                 uint64_t bin_size, uint64_t bin_count, uint32_t *counts, int mem, uint32_t n_parts,
                                 uint32_t part) {
    if (!table) return kt::fail(KT_ERR_ARG, "kt_cov_batch_part: null table");
    if (bin_size == 0) return kt::fail(KT_ERR_ARG, "kt_cov_batch_part: bin_size must be >= 1");
    if (bin_count == 0 || bin_count > 0xFFFFFFFFull) return kt::fail(KT_ERR_ARG, "kt_cov_batch_part: bin_count must be in 1..2^32-1");
    if (n_parts < 1 || part >= n_parts) return kt::fail(KT_ERR_ARG, "kt_cov_batch_part: need part < n_parts");
    if (mem != KT_MEM_HOST && mem != KT_MEM_DEVICE) return kt::fail(KT_ERR_ARG, "kt_cov_batch_part: bad mem");
    if (n_reads == 0) return KT_OK;
    if (!offsets || !counts) return kt::fail(KT_ERR_ARG, "kt_cov_batch_part: null buffer");
    kt_ctx *ctx = table->ctx;
    if (int rc = ctx->use()) return rc;
    if (int rc = table_ready(table)) return rc;
    uint64_t total = 0;
    if (int rc = total_bases_of(ctx, offsets, n_reads, mem, &total)) return rc;
    if (total && !bases) return kt::fail(KT_ERR_ARG, "kt_cov_batch_part: null bases");
    const uint64_t n_cells = n_reads * bin_count;
    if (mem == KT_MEM_DEVICE) return cov_counts(table, ctx, bases, offsets, n_reads, total, bin_size, bin_count, counts, n_parts, part);
    // host rows: this call's counts are made on the device from zero and added to the caller's
    const uint8_t *d_bases = bases;
    const uint64_t *d_offsets = offsets;
    if (total)
        if (int rc = stage_batch(ctx, bases, offsets, n_reads, &d_bases, &d_offsets)) return rc;
    if (int rc = ctx->s_aux1.reserve(n_cells * 4)) return rc;
    uint32_t *d_counts = (uint32_t *)ctx->s_aux1.p;
    KT_HIP(hipMemsetAsync(d_counts, 0, n_cells * 4, ctx->stream));
    if (int rc = cov_counts(table, ctx, d_bases, d_offsets, n_reads, total, bin_size, bin_count, d_counts, n_parts, part)) return rc;
    uint32_t *tmp = (uint32_t *)malloc(n_cells * 4);
    if (!tmp) return kt::fail(KT_ERR_NOMEM, "kt_cov_batch_part: host alloc");
    hipError_t e = hipMemcpyAsync(tmp, d_counts, n_cells * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess)
        for (uint64_t i = 0; i < n_cells; i++) counts[i] += tmp[i];
    free(tmp);
    if (e != hipSuccess) return kt::fail(KT_ERR_HIP, std::string("kt_cov_batch_part: ") + hipGetErrorString(e));
    return KT_OK;
}

extern "C" int kt_ctr_lookup(kt_ctr *table, const uint64_t *keys, uint64_t n, uint32_t *counts, int mem) {
    if (!table) return kt::fail(KT_ERR_ARG, "kt_ctr_lookup: null table");
    if (mem != KT_MEM_HOST && mem != KT_MEM_DEVICE) return kt::fail(KT_ERR_ARG, "kt_ctr_lookup: bad mem");
    if (n == 0) return KT_OK;
    if (!keys || !counts) return kt::fail(KT_ERR_ARG, "kt_ctr_lookup: null buffer");
    kt_ctx *ctx = table->ctx;
    if (int rc = ctx->use()) return rc;
    if (int rc = table_ready(table)) return rc;  // (a densely packed table gets its probing image first)
    const uint64_t *d_keys = keys;
    uint32_t *d_counts = counts;
    if (mem == KT_MEM_HOST) {
        if (int rc = ctx->s_aux1.reserve(n * 8)) return rc;
        if (int rc = ctx->s_aux2.reserve(n * 4)) return rc;
        KT_HIP(hipMemcpyAsync(ctx->s_aux1.p, keys, n * 8, hipMemcpyHostToDevice, ctx->stream));
        d_keys = (const uint64_t *)ctx->s_aux1.p;
        d_counts = (uint32_t *)ctx->s_aux2.p;
    }
    hipLaunchKernelGGL(lookup_kernel, dim3(grid_for(ctx, (n + BLOCK - 1) / BLOCK, 8)), dim3(BLOCK), 0, ctx->stream,
                       (const Slot *)table->slots, ktl::geom_of(table), d_keys, n, d_counts);
    KT_HIP(hipGetLastError());
    if (mem == KT_MEM_HOST) {
        KT_HIP(hipMemcpyAsync(counts, d_counts, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipStreamSynchronize(ctx->stream));
    }
    return KT_OK;
}

extern "C" int kt_cov_batch(kt_ctr *table, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                            uint64_t bin_size, uint64_t bin_count, int norm, int out_dtype, void *out, int mem) {
    if (!table) return kt::fail(KT_ERR_ARG, "kt_cov_batch: null table");
    if (bin_size == 0) return kt::fail(KT_ERR_ARG, "kt_cov_batch: bin_size must be >= 1");
    if (bin_count == 0 || bin_count > 0xFFFFFFFFull) return kt::fail(KT_ERR_ARG, "kt_cov_batch: bin_count must be in 1..2^32-1");
    if (out_dtype != KT_F64 && out_dtype != KT_F32 && out_dtype != KT_U32)
        return kt::fail(KT_ERR_ARG, "kt_cov_batch: bad out_dtype");
    if (out_dtype == KT_U32 && norm) return kt::fail(KT_ERR_ARG, "kt_cov_batch: KT_U32 output needs norm = 0");
    if (mem != KT_MEM_HOST && mem != KT_MEM_DEVICE) return kt::fail(KT_ERR_ARG, "kt_cov_batch: bad mem");
    if (n_reads == 0) return KT_OK;
    if (!offsets || !out) return kt::fail(KT_ERR_ARG, "kt_cov_batch: null buffer");
    if (table->n_owners > 1)
        return kt::fail(KT_ERR_ARG, "kt_cov_batch: the table is one shard of a sharded table - kt_cov_batch_part on every shard, summed");
    kt_ctx *ctx = table->ctx;
    if (int rc = ctx->use()) return rc;
    if (int rc = table_ready(table)) return rc;
    uint64_t total = 0;
    if (int rc = total_bases_of(ctx, offsets, n_reads, mem, &total)) return rc;
    if (total && !bases) return kt::fail(KT_ERR_ARG, "kt_cov_batch: null bases");

    const uint64_t n_cells = n_reads * bin_count;
    const size_t esz = out_dtype == KT_F64 ? 8 : 4;
    const uint8_t *d_bases = bases;
    const uint64_t *d_offsets = offsets;
    void *d_out = out;
    if (mem == KT_MEM_HOST) {
        if (total) {
            if (int rc = stage_batch(ctx, bases, offsets, n_reads, &d_bases, &d_offsets)) return rc;
        }
        if (int rc = ctx->s_out.reserve(n_cells * esz)) return rc;
        d_out = ctx->s_out.p;
    }
    // u32 bin counts: the output itself for KT_U32, otherwise scratch
    uint32_t *d_counts = (uint32_t *)d_out;
    if (out_dtype != KT_U32) {
        if (int rc = ctx->s_aux1.reserve(n_cells * 4)) return rc;
        d_counts = (uint32_t *)ctx->s_aux1.p;
    }
    KT_HIP(hipMemsetAsync(d_counts, 0, n_cells * 4, ctx->stream));
    if (int rc = cov_counts(table, ctx, d_bases, d_offsets, n_reads, total, bin_size, bin_count, d_counts, 1, 0)) return rc;
    const uint32_t fb = (uint32_t)((n_reads + BLOCK - 1) / BLOCK);
    if (out_dtype == KT_F64)
        hipLaunchKernelGGL(cov_finalize_kernel<double>, dim3(fb), dim3(BLOCK), 0, ctx->stream, d_counts, n_reads,
                           (uint32_t)bin_count, norm, (double *)d_out);
    else if (out_dtype == KT_F32)
        hipLaunchKernelGGL(cov_finalize_kernel<float>, dim3(fb), dim3(BLOCK), 0, ctx->stream, d_counts, n_reads,
                           (uint32_t)bin_count, norm, (float *)d_out);
    KT_HIP(hipGetLastError());
    if (mem == KT_MEM_HOST) {
        KT_HIP(hipMemcpyAsync(out, d_out, n_cells * esz, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipStreamSynchronize(ctx->stream));
    }
    return KT_OK;
}
