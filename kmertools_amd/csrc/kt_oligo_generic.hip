// kt_oligo_generic.hip - oligonucleotide histograms for k outside the CLI's 3..7 (k = 1, 2, 8..12).
//
// The reference's Python OligoComputer takes any ksize (pybindings/src/oligo.rs:22-31; the CLI alone restricts it,
// kmertools/src/args.rs:85), so a drop-in has to as well.  Rows of 4^k / 2 bins do not fit LDS beyond k = 7
// (k = 8: 132 KB of counters), so this path counts in global memory: the segment front-end of the counting kernels
// yields every k-mer with the index of its last base, the read is found by a binary search over the reads that touch
// the segment, and the bin's u32 counter is bumped with a global atomic; a second kernel turns counters into the
// output element type (one IEEE division per bin, like the reference's `vec[i] /= max(1, total)`).  Reads are taken in
// slabs so that the counters stay below 1 GiB.  It is the uncommon path: correct first, then as fast as atomics allow.
#include <vector>

#include "kt_internal.hpp"
#include "kt_launch.hpp"
#include "kt_segment.hpp"

namespace {

using ktseg::SegArgs;
using ktseg::SegShared;
constexpr int BLOCK = ktseg::BLOCK;

template <bool CANON>
__global__ __launch_bounds__(BLOCK) void oligo_generic_count(SegArgs a, const uint32_t *__restrict__ lut, uint64_t bins,
                                                             uint32_t *__restrict__ counts,
                                                             uint32_t *__restrict__ totals) {
    __shared__ SegShared sm;
    for (uint64_t g = blockIdx.x; g < a.n_seg; g += gridDim.x) {
        const uint64_t r_first = a.seg_first[g];
        const uint64_t r_lo = r_first ? r_first - 1 : 0;  // offsets[r_lo] <= g * SEG
        ktseg::for_each_kmer(a, g, sm, [&](uint64_t f, uint64_t r, uint64_t end) {
            // the read that holds base `end`: the last read with offsets[read] <= end (empty reads share an offset
            // with their successor and are stepped over that way)
            uint64_t lo = r_lo, hi = a.n_reads;
            while (hi - lo > 1) {
                const uint64_t mid = (lo + hi) >> 1;
                if (a.offsets[mid] <= end) lo = mid; else hi = mid;
            }
            const uint64_t m = f < r ? f : r;
            const uint64_t bin = CANON ? (uint64_t)lut[m] : f;
            atomicAdd(&counts[lo * bins + bin], 1u);
            atomicAdd(&totals[lo], 1u);
        });
    }
}

template <class T>
__global__ __launch_bounds__(BLOCK) void oligo_generic_finish(const uint32_t *__restrict__ counts,
                                                              const uint32_t *__restrict__ totals, uint64_t n_reads,
                                                              uint64_t bins, int norm, uint32_t total_step,
                                                              T *__restrict__ out) {
    const uint64_t cells = n_reads * bins;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < cells; i += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t r = i / bins;
        const double c = (double)counts[i];
        if (norm) {
            const double t = (double)((uint64_t)totals[r] * total_step);
            out[i] = (T)(c / (t > 1.0 ? t : 1.0));  // f64::max(1, total), oligo.rs:255-257
        } else {
            out[i] = (T)c;
        }
    }
}

__global__ void rebase_offsets_kernel(const uint64_t *__restrict__ offsets, uint64_t r0, uint64_t n, uint64_t *__restrict__ out) {
    const uint64_t base = offsets[r0];
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = offsets[r0 + i] - base;
}

}  // namespace

int kt_ctx::canon_lut32(int k, const uint32_t **out) {
    if (k < 1 || k > 12) return kt::fail(KT_ERR_ARG, "canon_lut32: k out of range");
    if (!lut32_dev[k]) {
        const size_t n = (size_t)1 << (2 * k);
        std::vector<uint32_t> h(n);
        uint32_t rank = 0;
        for (uint64_t km = 0; km < n; km++)
            if (km <= ktd::rev_comp(km, k)) h[km] = rank++;
        for (uint64_t km = 0; km < n; km++) {
            const uint64_t rc = ktd::rev_comp(km, k);
            if (km > rc) h[km] = h[rc];
        }
        uint32_t *d = nullptr;
        KT_HIP(hipMalloc((void **)&d, n * sizeof(uint32_t)));
        hipError_t e = hipMemcpy(d, h.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d);
            return kt::fail(KT_ERR_HIP, std::string("hipMemcpy lut32: ") + hipGetErrorString(e));
        }
        lut32_dev[k] = d;
    }
    *out = lut32_dev[k];
    return KT_OK;
}

// device-resident CSR batch -> device-resident rows; enqueued on the ctx stream
int kt_oligo_generic_launch(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int k,
                            int count_min, int norm, int total_step, int dt, void *out) {
    uint64_t bins = 0;
    if (int rc = kt_bins(k, count_min, &bins)) return rc;
    const uint32_t *lut = nullptr;
    if (count_min) {
        if (int rc = ctx->canon_lut32(k, &lut)) return rc;
    }
    const size_t esz = dt == KT_F64 ? 8 : 4;
    uint64_t slab = (1ull << 28) / bins;  // <= 1 GiB of u32 counters at a time
    if (slab < 1) slab = 1;
    if (slab > n_reads) slab = n_reads;
    if (int rc = ctx->s_aux1.reserve(slab * bins * 4)) return rc;
    if (int rc = ctx->s_aux2.reserve(slab * 4 + (slab + 1) * 8 + 64)) return rc;
    uint32_t *counts = (uint32_t *)ctx->s_aux1.p;
    uint32_t *totals = (uint32_t *)ctx->s_aux2.p;
    uint64_t *sub_offsets = (uint64_t *)((char *)ctx->s_aux2.p + ((slab * 4 + 63) & ~(size_t)63));
    for (uint64_t r0 = 0; r0 < n_reads; r0 += slab) {
        const uint64_t nr = n_reads - r0 < slab ? n_reads - r0 : slab;
        uint64_t b0 = 0, b1 = 0;
        KT_HIP(hipMemcpyAsync(&b0, offsets + r0, 8, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipMemcpyAsync(&b1, offsets + r0 + nr, 8, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipStreamSynchronize(ctx->stream));
        KT_HIP(hipMemsetAsync(counts, 0, nr * bins * 4, ctx->stream));
        KT_HIP(hipMemsetAsync(totals, 0, nr * 4, ctx->stream));
        if (b1 > b0) {
            hipLaunchKernelGGL(rebase_offsets_kernel, dim3((uint32_t)((nr + 256) / 256)), dim3(256), 0, ctx->stream, offsets, r0,
                               nr, sub_offsets);
            SegArgs a;
            if (int rc = ktl::make_seg_args(ctx, bases + b0, sub_offsets, nr, b1 - b0, k, &a)) return rc;
            if (count_min)
                hipLaunchKernelGGL(oligo_generic_count<true>, dim3(ktl::grid_for(ctx, a.n_seg, 8)), dim3(BLOCK), 0, ctx->stream,
                                   a, lut, bins, counts, totals);
            else
                hipLaunchKernelGGL(oligo_generic_count<false>, dim3(ktl::grid_for(ctx, a.n_seg, 8)), dim3(BLOCK), 0, ctx->stream,
                                   a, lut, bins, counts, totals);
        }
        char *dst = (char *)out + r0 * bins * esz;
        const uint32_t grid = ktl::grid_for(ctx, (nr * bins + BLOCK - 1) / BLOCK, 16);
        if (dt == KT_F64)
            hipLaunchKernelGGL(oligo_generic_finish<double>, dim3(grid), dim3(BLOCK), 0, ctx->stream, counts, totals, nr, bins,
                               norm, (uint32_t)total_step, (double *)dst);
        else if (dt == KT_F32)
            hipLaunchKernelGGL(oligo_generic_finish<float>, dim3(grid), dim3(BLOCK), 0, ctx->stream, counts, totals, nr, bins,
                               norm, (uint32_t)total_step, (float *)dst);
        else
            hipLaunchKernelGGL(oligo_generic_finish<uint32_t>, dim3(grid), dim3(BLOCK), 0, ctx->stream, counts, totals, nr, bins,
                               0, 1u, (uint32_t *)dst);
        KT_HIP(hipGetLastError());
    }
    return KT_OK;
}
