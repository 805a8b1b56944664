// kt_oligo.hip - per-read oligonucleotide-frequency histograms on gfx950.
//
// Replaces OligoComputer::vectorise_one (reference composition/src/oligo.rs:231-259),
// OligoCgrComputer::seq_to_kmer (composition/src/oligocgr.rs:145-163) and the python
// binding's copy (pybindings/src/oligo.rs:39-69) for a whole CSR batch of reads.
//
// Shape of the problem (k=4, 150-bp reads, f64 rows): 150 B in, 1088 B out per read -
// a streaming-store kernel.  HBM-bound; no MFMA (integer histogram).
//
// One 256-thread workgroup owns a *tile* of R consecutive reads:
//   stage     the tile's bytes are contiguous in `bases`; all 256 lanes copy them to LDS
//             with 16-byte coalesced loads (one wait per tile instead of one per read)
//   positions work items are (read, 64-base chunk); chunk c of read r goes to wave
//             (c + r) & 3, so short reads fill all four waves and a long read is spread
//             over the workgroup.  Lane l holds base c*S + l, S = 64-(k-1); the k-1
//             predecessors arrive by DPP wave_shr:1 (no LDS traffic); the k-mer is
//             fwd(p) = sum code[p-j]*4^j - position-parallel, identical to the
//             reference's rolling value.  bin = lut[fwd] (rank of the
//             canonical form) or fwd; one ds_add_u32 into the read's LDS histogram.
//   output    the R x bins counts are one contiguous block of the output matrix; the
//             256 lanes stream it out as 16-byte stores with the normalising division
//             fused, and clear the LDS histogram behind them.
#include "kt_device.hpp"
#include "kt_internal.hpp"

namespace {

constexpr int BLOCK = 256;
constexpr int NWAVES = BLOCK / 64;

template <int DT>
struct OutVec;
template <>
struct OutVec<KT_F64> {
    static constexpr int VEC = 2;
    using type = double2;
};
template <>
struct OutVec<KT_F32> {
    static constexpr int VEC = 4;
    using type = float4;
};
template <>
struct OutVec<KT_U32> {
    static constexpr int VEC = 4;
    using type = uint4;
};

struct OligoArgs {
    const uint8_t *bases;
    const uint64_t *offsets;
    uint64_t n_reads;
    const uint16_t *lut;  // device, 4^k entries (canonical mode)
    void *out;
    uint32_t bins;
    uint32_t R;            // reads per tile
    uint32_t stage_bytes;  // LDS bytes reserved for the tile's bases
    uint32_t norm;
    uint32_t total_step;
    uint32_t vec_per_row;  // bins / VEC
    uint32_t vec_magic;    // ceil(2^32 / vec_per_row)
};

template <int K, bool CANON, int DT>
__global__ __launch_bounds__(BLOCK) void oligo_tile_kernel(OligoArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int VEC = OutVec<DT>::VEC;
    using vec_t = typename OutVec<DT>::type;
    constexpr uint32_t S = 64 - (K - 1);
    constexpr uint32_t NLUT = CANON ? (1u << (2 * K)) : 0u;

    const uint32_t R = a.R, bins = a.bins;
    // LDS carve (all offsets multiples of 16)
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);                      // R * bins
    uint32_t off = R * bins * 4;
    uint32_t *totals = reinterpret_cast<uint32_t *>(smem + off);              // 2 * R (double buffered)
    off += ((2 * R * 4 + 15) & ~15u);
    uint64_t *roff = reinterpret_cast<uint64_t *>(smem + off);                // R + 1
    off += (((R + 1) * 8 + 15) & ~15u);
    uint16_t *lut = reinterpret_cast<uint16_t *>(smem + off);                 // 4^K
    off += ((NLUT * 2 + 15) & ~15u);
    unsigned char *stage = smem + off;                                        // stage_bytes

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    for (uint32_t i = tid; i < R * bins; i += BLOCK) hist[i] = 0;
    for (uint32_t i = tid; i < 2 * R; i += BLOCK) totals[i] = 0;
    if (CANON)
        for (uint32_t i = tid; i < NLUT; i += BLOCK) lut[i] = a.lut[i];

    const uint64_t n_tiles = (a.n_reads + R - 1) / R;
    const uint64_t total_bytes = a.offsets[a.n_reads];
    const uintptr_t base_addr = reinterpret_cast<uintptr_t>(a.bases);
    uint32_t parity = 0;

    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, parity ^= 1u) {
        const uint64_t r0 = tile * R;
        const uint32_t nr = (uint32_t)((a.n_reads - r0) < R ? (a.n_reads - r0) : R);
        const uint64_t off0 = a.offsets[r0], off1 = a.offsets[r0 + nr];
        for (uint32_t i = tid; i <= nr; i += BLOCK) roff[i] = a.offsets[r0 + i];

        // ---- stage the tile's bases ------------------------------------------------
        const uintptr_t addr0 = base_addr + off0;
        const uintptr_t al0 = addr0 & ~(uintptr_t)15;
        const uint32_t sh = (uint32_t)(addr0 - al0);
        const uint64_t span = (off1 - off0) + sh;
        const bool staged = span + 16 <= a.stage_bytes;
        if (staged) {
            const uint32_t n16 = (uint32_t)((span + 15) >> 4);
            for (uint32_t i = tid; i < n16; i += BLOCK) {
                const uintptr_t p = al0 + 16ull * i;
                uint4 v;
                if (p >= base_addr && p + 16 <= base_addr + total_bytes) {
                    v = *reinterpret_cast<const uint4 *>(p);
                } else {  // first/last 16 bytes of the whole buffer: stay inside it
                    unsigned char b[16];
                    for (int j = 0; j < 16; j++) {
                        const uintptr_t q = p + j;
                        b[j] = (q >= base_addr && q < base_addr + total_bytes)
                                   ? *reinterpret_cast<const unsigned char *>(q)
                                   : (unsigned char)'N';
                    }
                    __builtin_memcpy(&v, b, 16);
                }
                *reinterpret_cast<uint4 *>(stage + 16u * i) = v;
            }
        }
        __syncthreads();

        // ---- positions -> LDS histograms --------------------------------------------
        uint32_t *tot = totals + parity * R;
        for (uint32_t r = 0; r < nr; r++) {
            const uint64_t rs = roff[r];
            const uint64_t L64 = roff[r + 1] - rs;
            const uint32_t L = (uint32_t)(L64 > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : L64);
            if (L < (uint32_t)K) continue;
            const uint32_t nch = (L - (K - 1) + S - 1) / S;
            const unsigned char *src_l = stage + sh + (uint32_t)(rs - off0);
            const uint8_t *src_g = a.bases + rs;
            uint32_t *h = hist + r * bins;
            uint32_t cnt = 0;
            for (uint32_t c = (wave - r) & (NWAVES - 1); c < nch; c += NWAVES) {
                const uint32_t p = c * S + lane;
                uint32_t e = 4u;
                if (p < L) e = ktd::nt4(staged ? src_l[p] : src_g[p]);
                uint32_t f = e & 3u, bad = e >> 2, x = e;
#pragma unroll
                for (int j = 1; j < K; j++) {
                    x = ktd::wave_shr1(x, 4u);
                    f |= (x & 3u) << (2 * j);
                    bad |= x >> 2;
                }
                const bool emit = (lane >= (uint32_t)(K - 1)) && (bad == 0);
                if (emit) {
                    const uint32_t bin = CANON ? (uint32_t)lut[f] : f;
                    atomicAdd(&h[bin], 1u);
                }
                cnt += (uint32_t)__popcll(__ballot(emit));
            }
            if (lane == 0 && cnt) atomicAdd(&tot[r], cnt);
        }
        __syncthreads();

        // ---- stream the tile's rows out, clearing the histogram behind ----------------
        const uint32_t nvec = nr * a.vec_per_row;
        vec_t *dst = reinterpret_cast<vec_t *>(a.out) + r0 * a.vec_per_row;
        for (uint32_t v = tid; v < nvec; v += BLOCK) {
            const uint32_t r = __umulhi(v, a.vec_magic);
            uint32_t *hp = hist + v * VEC;
            vec_t o;
            if constexpr (DT == KT_F64) {
                const uint2 c = *reinterpret_cast<uint2 *>(hp);
                *reinterpret_cast<uint2 *>(hp) = make_uint2(0, 0);
                double d = 1.0;
                if (a.norm) {
                    const double t = (double)((uint64_t)tot[r] * a.total_step);
                    d = t > 1.0 ? t : 1.0;
                }
                o.x = (double)c.x / d;
                o.y = (double)c.y / d;
            } else {
                const uint4 c = *reinterpret_cast<uint4 *>(hp);
                *reinterpret_cast<uint4 *>(hp) = make_uint4(0, 0, 0, 0);
                if constexpr (DT == KT_F32) {
                    float d = 1.0f;
                    if (a.norm) {
                        const float t = (float)((uint64_t)tot[r] * a.total_step);
                        d = t > 1.0f ? t : 1.0f;
                    }
                    o.x = (float)c.x / d;
                    o.y = (float)c.y / d;
                    o.z = (float)c.z / d;
                    o.w = (float)c.w / d;
                } else {
                    o = c;
                }
            }
            dst[v] = o;
        }
        // the other totals buffer was last read one tile ago: safe to clear now
        for (uint32_t i = tid; i < R; i += BLOCK) totals[(parity ^ 1u) * R + i] = 0;
        // the next tile's first barrier orders these LDS writes before its positions phase
    }
}

using kern_t = void (*)(OligoArgs);

template <int K>
kern_t pick(int count_min, int dt) {
    if (count_min) {
        switch (dt) {
            case KT_F64: return oligo_tile_kernel<K, true, KT_F64>;
            case KT_F32: return oligo_tile_kernel<K, true, KT_F32>;
            default: return oligo_tile_kernel<K, true, KT_U32>;
        }
    }
    switch (dt) {
        case KT_F64: return oligo_tile_kernel<K, false, KT_F64>;
        case KT_F32: return oligo_tile_kernel<K, false, KT_F32>;
        default: return oligo_tile_kernel<K, false, KT_U32>;
    }
}

uint32_t env_u32(const char *name, uint32_t dflt) {
    const char *s = getenv(name);
    if (!s || !*s) return dflt;
    long v = strtol(s, nullptr, 10);
    return v > 0 ? (uint32_t)v : dflt;
}

}  // namespace

// Enqueue the histogram kernel for device-resident inputs/outputs.
static int oligo_launch(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                        int k, int count_min, int norm, int total_step, int dt, void *out) {
    uint64_t bins64 = 0;
    kt_bins(k, count_min, &bins64);
    const uint32_t bins = (uint32_t)bins64;
    const int VEC = dt == KT_F64 ? 2 : 4;

    OligoArgs a{};
    a.bases = bases;
    a.offsets = offsets;
    a.n_reads = n_reads;
    a.lut = nullptr;
    if (count_min) {
        if (int rc = ctx->canon_lut(k, &a.lut)) return rc;
    }
    a.out = out;
    a.bins = bins;
    a.norm = (uint32_t)(norm != 0);
    a.total_step = (uint32_t)total_step;
    a.vec_per_row = bins / VEC;
    a.vec_magic = (uint32_t)((0x100000000ull + a.vec_per_row - 1) / a.vec_per_row);

    // reads per tile: ~18 KB of LDS histogram, at most 64 reads; the flat output index
    // v < R * vec_per_row must keep the magic division exact: v * vec_per_row < 2^32.
    uint32_t R = 18432u / (bins * 4u);
    if (R < 1) R = 1;
    if (R > 64) R = 64;
    R = env_u32("KT_OLIGO_R", R);
    while (R > 1 && (uint64_t)R * a.vec_per_row * a.vec_per_row >= 0x100000000ull) R--;
    a.R = R;
    a.stage_bytes = ((R * 192u + 47u) & ~15u);
    if (a.stage_bytes > 24576u) a.stage_bytes = 24576u;
    a.stage_bytes = env_u32("KT_OLIGO_STAGE", a.stage_bytes) & ~15u;

    size_t lds = (size_t)R * bins * 4;
    lds += ((2 * R * 4 + 15) & ~15u);
    lds += (((R + 1) * 8 + 15) & ~15u);
    lds += count_min ? ((((size_t)1 << (2 * k)) * 2 + 15) & ~(size_t)15) : 0;
    lds += a.stage_bytes;
    if (lds > 160 * 1024) return kt::fail(KT_ERR_ARG, "kt_oligo_batch: tile does not fit in LDS");

    kern_t kern = nullptr;
    switch (k) {
        case 3: kern = pick<3>(count_min, dt); break;
        case 4: kern = pick<4>(count_min, dt); break;
        case 5: kern = pick<5>(count_min, dt); break;
        case 6: kern = pick<6>(count_min, dt); break;
        case 7: kern = pick<7>(count_min, dt); break;
        default: return kt::fail(KT_ERR_ARG, "kt_oligo_batch: k must be in 3..7");
    }
    if (lds > 64 * 1024)
        KT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

    const uint64_t n_tiles = (n_reads + R - 1) / R;
    uint32_t per_cu = (uint32_t)((160 * 1024) / lds);
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    uint64_t grid = (uint64_t)ctx->n_cu * per_cu * env_u32("KT_OLIGO_OVERSUB", 1);
    if (grid > n_tiles) grid = n_tiles;
    if (grid == 0) return KT_OK;
    hipLaunchKernelGGL(kern, dim3((uint32_t)grid), dim3(BLOCK), lds, ctx->stream, a);
    KT_HIP(hipGetLastError());
    return KT_OK;
}

extern "C" int kt_oligo_batch(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets,
                              uint64_t n_reads, int k, int count_min, int norm, int total_step,
                              int out_dtype, void *out, int mem) {
    if (!ctx) return kt::fail(KT_ERR_ARG, "kt_oligo_batch: null ctx");
    if (k < 3 || k > 7) return kt::fail(KT_ERR_ARG, "kt_oligo_batch: k must be in 3..7");
    if (out_dtype != KT_F64 && out_dtype != KT_F32 && out_dtype != KT_U32)
        return kt::fail(KT_ERR_ARG, "kt_oligo_batch: unknown out_dtype");
    if (out_dtype == KT_U32 && norm) return kt::fail(KT_ERR_ARG, "kt_oligo_batch: KT_U32 output needs norm = 0");
    if (total_step < 1 || total_step > 2) return kt::fail(KT_ERR_ARG, "kt_oligo_batch: total_step must be 1 or 2");
    if (mem != KT_MEM_HOST && mem != KT_MEM_DEVICE) return kt::fail(KT_ERR_ARG, "kt_oligo_batch: bad mem flag");
    if (n_reads == 0) return KT_OK;
    if (!offsets || !out) return kt::fail(KT_ERR_ARG, "kt_oligo_batch: null pointer");
    if (int rc = ctx->use()) return rc;

    uint64_t bins = 0;
    kt_bins(k, count_min, &bins);
    const size_t esz = out_dtype == KT_F64 ? 8 : 4;

    if (mem == KT_MEM_DEVICE) {
        if (!bases) return kt::fail(KT_ERR_ARG, "kt_oligo_batch: null bases");
        if (reinterpret_cast<uintptr_t>(out) & 15u)
            return kt::fail(KT_ERR_ARG, "kt_oligo_batch: device output must be 16-byte aligned");
        return oligo_launch(ctx, bases, offsets, n_reads, k, count_min, norm, total_step, out_dtype, out);
    }

    // host buffers: stage through ctx scratch, in slabs so the output scratch stays bounded
    const uint64_t total = offsets[n_reads];
    if (total && !bases) return kt::fail(KT_ERR_ARG, "kt_oligo_batch: null bases");
    const uint64_t row_bytes = bins * esz;
    uint64_t slab = (1ull << 30) / row_bytes;  // ~1 GiB of output per slab
    if (slab < 1) slab = 1;
    for (uint64_t r0 = 0; r0 < n_reads; r0 += slab) {
        const uint64_t nr = (n_reads - r0) < slab ? (n_reads - r0) : slab;
        const uint64_t b0 = offsets[r0], b1 = offsets[r0 + nr];
        if (int rc = ctx->s_bases.reserve(b1 - b0 + 16)) return rc;
        if (int rc = ctx->s_offsets.reserve((nr + 1) * 8)) return rc;
        if (int rc = ctx->s_out.reserve(nr * row_bytes)) return rc;
        // rebase offsets to the slab
        uint64_t *tmp = (uint64_t *)malloc((nr + 1) * 8);
        if (!tmp) return kt::fail(KT_ERR_NOMEM, "kt_oligo_batch: host alloc");
        for (uint64_t i = 0; i <= nr; i++) tmp[i] = offsets[r0 + i] - b0;
        hipError_t e = hipSuccess;
        if (b1 > b0) e = hipMemcpyAsync(ctx->s_bases.p, bases + b0, b1 - b0, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(ctx->s_offsets.p, tmp, (nr + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
        int rc = KT_OK;
        if (e == hipSuccess)
            rc = oligo_launch(ctx, (const uint8_t *)ctx->s_bases.p, (const uint64_t *)ctx->s_offsets.p, nr, k,
                              count_min, norm, total_step, out_dtype, ctx->s_out.p);
        if (e == hipSuccess && rc == KT_OK)
            e = hipMemcpyAsync((char *)out + r0 * row_bytes, ctx->s_out.p, nr * row_bytes,
                               hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && rc == KT_OK) e = hipStreamSynchronize(ctx->stream);
        free(tmp);
        if (rc != KT_OK) return rc;
        if (e != hipSuccess) return kt::fail(KT_ERR_HIP, std::string("kt_oligo_batch: ") + hipGetErrorString(e));
    }
    return KT_OK;
}
