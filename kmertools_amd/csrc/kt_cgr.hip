// kt_cgr.hip - whole-sequence Chaos Game Representation (comp cgr without -k).
//
// Replaces CgrComputer::vectorise_one (reference composition/src/cgr.rs:127-144 and the copy in
// pybindings/src/cgr.rs:38-54): marker = (V/2, V/2); for every base marker = (corner + marker) / 2,
// emitted as one (x, y) pair of f64 per base.  16 bytes out per byte in: HBM-store bound, no MFMA.
//
// The walk is a serial f64 recurrence with a rounding at every step, and results must be
// bit-identical, so it cannot be re-associated into a scan.  It can still start anywhere:
// the step m -> RN(c + m) / 2 is monotone in m, so two walks started W bases early from the
// lowest (0,0) and highest (V,V) markers bracket the true marker, and once they coincide the
// true marker is known exactly.  The gap halves per step, so 64 bases almost always suffice
// (the window is widened, up to the read start, when they do not - e.g. inside a long poly-A).
// That turns one long dependent chain per read into independent CH-base chunks:
//   wave   = 64 consecutive chunks of the concatenated batch; the read starts that fall in them are
//            scattered into an LDS bit mask first (found through seg_first[]), so the walks never
//            track read ids or touch the offsets array: a set bit puts the marker back on the centre
//   lane   = one chunk: recovers the marker at its first base (bracketing walk over the 64 bases
//            before it), then walks its CH bases
//   store  = every 8 steps the wave's 64 x 8 points go through LDS so that each store
//            instruction writes 128-byte runs instead of 64 scattered 16-byte points.  Small chunks
//            keep a wave's 64 output streams close together (HBM wants the neighbouring lines soon):
//            10 M x 150 bp takes 4.8 ms at CH = 64..128, 5.5 ms at CH = 256; below 64 the 64-base
//            bracket per chunk makes the walk itself the bound (7.0 ms at 32, 9.5 ms at 16).
// A byte outside ACGTUacgtu (cgr.rs:19-30 - raw codes 0..3 are NOT letters here) is an error for
// the whole call, as in the reference: the lowest offending index is reported.
#include "kt_internal.hpp"
#include "kt_launch.hpp"
#include "kt_segment.hpp"

#ifndef KT_CGR_CH
#define KT_CGR_CH 128
#endif
#ifndef KT_CGR_GROUP
#define KT_CGR_GROUP 8
#endif
#ifndef KT_CGR_BLOCK
#define KT_CGR_BLOCK 256
#endif
#ifndef KT_CGR_DEBUG
#define KT_CGR_DEBUG 0  // timing experiments only: 1 = no bracketing start, 2 = no stores
#endif

namespace {

constexpr int BLOCK = KT_CGR_BLOCK;
constexpr int WAVES = BLOCK / 64;
constexpr uint32_t CH = KT_CGR_CH;           // bases per lane
constexpr uint32_t SPAN = 64 * CH;           // bases per wave
constexpr uint32_t GROUP = KT_CGR_GROUP;     // steps between LDS transposes
constexpr uint32_t PITCH = GROUP + 1;        // double2 per lane row (+1: spreads the rows over the banks)
constexpr uint32_t WARM = 64;                // bracketing window
constexpr uint32_t PRE = CH > WARM ? CH : WARM;  // bases before the span that the read-start mask covers
constexpr uint32_t MASK_WORDS = (PRE + SPAN) / 32;
static_assert(CH % 8 == 0 && GROUP % 8 == 0 && CH % GROUP == 0 && SPAN % 32 == 0 && ktseg::SEG % CH == 0, "chunking");

// Corners (cgr.rs:19-30) from two bits of the letter: (b >> 1) & 3 is A=0 C=1 T/U=2 G=3 in either case,
// so x = V for T/U/G (bit 2 of the byte) and y = V for C/G (bit 1).  Returns the halved corner as the
// pre-halved vecsize ANDed with a sign-extended bit.
__device__ __forceinline__ double half_corner(uint32_t word, uint32_t bit, double hv) {
    const int32_t m = __builtin_amdgcn_sbfe(word, bit, 1);  // 0 or -1
    const uint64_t bits = (uint64_t)__double_as_longlong(hv) & (((uint64_t)(uint32_t)m << 32) | (uint32_t)m);
    return __longlong_as_double((long long)bits);
}

// is b one of ACGTUacgtu
__device__ __forceinline__ bool is_letter(uint32_t b) {
    const uint32_t l = b | 0x20u;
    return (l == 'a') | (l == 'c') | (l == 'g') | ((l & 0xFEu) == 't');
}

// 8 bases at p (any alignment); bytes past `total` read as 'A'
__device__ __forceinline__ uint2 load8(const uint8_t *bases, uint64_t p, uint64_t total) {
    uint2 w;
    if (p + 8 <= total) {
        __builtin_memcpy(&w, bases + p, 8);
    } else {
        uint64_t t = 0;
        for (uint32_t j = 0; j < 8; j++) t |= (uint64_t)(p + j < total ? bases[p + j] : (uint8_t)'A') << (8 * j);
        w = make_uint2((uint32_t)t, (uint32_t)(t >> 32));
    }
    return w;
}

// marker = (corner + marker) / 2.0 (cgr.rs:134-137) as one fused multiply-add per coordinate on the
// pre-halved corner: RN(c/2 + m/2) == RN(c + m) / 2 because scaling by two commutes with rounding (the
// result is either >= 1/2 or, for corner 0, a plain halving of m), so the bits are the reference's.
__device__ __forceinline__ void step(double &mx, double &my, double hx, double hy) {
    mx = __builtin_fma(mx, 0.5, hx);
    my = __builtin_fma(my, 0.5, hy);
}

// Slow path of the chunk start: the 64-base bracket did not close (e.g. inside a long single-letter
// run).  Finds the read, then widens the window until the two bounds meet or it reaches the read start.
__device__ __noinline__ double2 marker_slow(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, double v,
                                            uint64_t s) {
    const double hv = v * 0.5;
    uint64_t lo = 0, hi = n_reads;  // last r with offsets[r] <= s
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= s) lo = mid; else hi = mid;
    }
    const uint64_t rstart = offsets[lo];
    for (uint64_t back = 4 * WARM;; back *= 4) {
        if (s - rstart <= back) {  // the window reaches the read start: the walk is the reference's own
            double mx = hv, my = hv;
            for (uint64_t i = rstart; i < s; i++) {
                const uint32_t b = bases[i];
                step(mx, my, half_corner(b, 2, hv), half_corner(b, 1, hv));
            }
            return make_double2(mx, my);
        }
        double lx = 0.0, ly = 0.0, ux = v, uy = v;
        for (uint64_t i = s - back; i < s; i++) {
            const uint32_t b = bases[i];
            const double hx = half_corner(b, 2, hv), hy = half_corner(b, 1, hv);
            step(lx, ly, hx, hy);
            step(ux, uy, hx, hy);
        }
        if (lx == ux && ly == uy) return make_double2(lx, ly);
    }
}

// The staging rows and read-start masks are private to a wave, and a wave's LDS instructions execute in
// order, so between "every lane wrote" and "every lane reads other lanes' data" it is enough to stop the
// compiler from moving LDS accesses across the point: no s_barrier, no wait for outstanding global stores.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Pointers are direct kernel arguments (not members of a by-value struct that a non-inlined callee takes by
// reference): that keeps them in the global address space.  As generic pointers every access became a
// FLAT instruction, which counts against lgkmcnt - so each wait for the LDS transpose also waited for all
// outstanding stores, and walk and stores ran back to back (7.3 ms) instead of overlapped.
struct CgrShape {
    uint64_t n_reads, total;
    double vecsize;
};

__global__ __launch_bounds__(BLOCK) void cgr_kernel(const uint8_t *__restrict__ g_bases, const uint64_t *__restrict__ g_offsets,
                                                    const uint64_t *__restrict__ g_seg_first, CgrShape a,
                                                    double2 *__restrict__ g_out, unsigned long long *__restrict__ g_bad) {
    __shared__ double2 stage[WAVES][64 * PITCH];
    __shared__ uint32_t starts[WAVES][MASK_WORDS];  // bit per base of [span0 - PRE, span0 + SPAN): a read starts here
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    double2 *row = &stage[wave][lane * PITCH];
    uint32_t *st_mask = starts[wave];
    const double v = a.vecsize, hv = v * 0.5;

    const uint64_t span0 = ((uint64_t)blockIdx.x * WAVES + wave) * SPAN;  // first base of this wave
    if (span0 >= a.total) return;  // whole wave; nothing below is block-synchronous
    const uint32_t rel_s = PRE + lane * CH;  // bit index of this lane's first base
    const uint64_t s = span0 + (uint64_t)lane * CH;
    const bool live = s < a.total;

    // ---- read starts of the span (and of the PRE bases before it) as a bit mask ----
    for (uint32_t i = lane; i < MASK_WORDS; i += 64) st_mask[i] = 0;
    wave_sync();
    {
        const uint64_t lo_pos = span0 >= PRE ? span0 - PRE : 0;
        const uint64_t bias = span0 >= PRE ? 0 : PRE - span0;  // bit index of base lo_pos
        const uint64_t end = span0 + SPAN;
        for (uint64_t r = g_seg_first[lo_pos / ktseg::SEG] + lane; r < a.n_reads; r += 64) {
            const uint64_t o = g_offsets[r];
            if (o >= end) break;
            if (o >= lo_pos) {
                const uint32_t rel = (uint32_t)(o - lo_pos + bias);
                atomicOr(&st_mask[rel >> 5], 1u << (rel & 31u));
            }
        }
    }
    wave_sync();
    // the 8 mask bits of bases [rel, rel + 8), rel a multiple of 8
    auto mask8 = [&](uint32_t rel) { return (st_mask[rel >> 5] >> (rel & 31u)) & 0xFFu; };

    // ---- marker before base s ----
    double mx = hv, my = hv;
    if (live && !(mask8(rel_s) & 1u) && !(KT_CGR_DEBUG & 1)) {
        // the WARM bases before s: two walks from the extreme markers; a read start inside the window
        // puts both on the centre, otherwise they close in on each other by a factor two per base
        double lx = 0.0, ly = 0.0, ux = v, uy = v;
#pragma unroll 1
        for (uint32_t j8 = s >= WARM ? 0 : WARM - (uint32_t)s; j8 < WARM; j8 += 8) {  // never before base 0 (a read start)
            const uint2 w = load8(g_bases, s - WARM + j8, a.total);
            const uint32_t bm = mask8(rel_s - WARM + j8);
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                if ((bm >> j) & 1u) lx = ly = ux = uy = hv;
                const uint32_t word = j < 4 ? w.x : w.y;
                const double hx = half_corner(word, 8 * (j & 3u) + 2, hv), hy = half_corner(word, 8 * (j & 3u) + 1, hv);
                step(lx, ly, hx, hy);
                step(ux, uy, hx, hy);
            }
        }
        mx = lx;
        my = ly;
        if (lx != ux || ly != uy) {
            const double2 m = marker_slow(g_bases, g_offsets, a.n_reads, v, s);
            mx = m.x;
            my = m.y;
        }
    }

    // ---- the chunk: GROUP serial steps, then a transposed store ----
    uint64_t bad = ~0ull;
#pragma unroll 1
    for (uint32_t st = 0; st < CH; st += GROUP) {
        const uint64_t p0 = s + st;
        if (p0 < a.total) {
#pragma unroll
            for (uint32_t j8 = 0; j8 < GROUP; j8 += 8) {
                const uint2 w = load8(g_bases, p0 + j8, a.total);
                const uint32_t bm = mask8(rel_s + st + j8);
                uint32_t inv = 0;
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) {
                    if ((bm >> j) & 1u) mx = my = hv;  // a read starts here: back to the centre
                    const uint32_t word = j < 4 ? w.x : w.y;
                    const uint32_t sh = 8 * (j & 3u);
                    inv |= (is_letter((word >> sh) & 0xFFu) ? 0u : 1u) << j;
                    step(mx, my, half_corner(word, sh + 2, hv), half_corner(word, sh + 1, hv));
                    row[j8 + j] = make_double2(mx, my);
                }
                if (inv) {
                    const uint64_t p = p0 + j8 + (uint32_t)__builtin_ctz(inv);
                    if (p < a.total && p < bad) bad = p;
                }
            }
        }
        wave_sync();
        if (!(KT_CGR_DEBUG & 2)) {
#pragma unroll
            for (uint32_t q = 0; q < GROUP; q++) {  // consecutive lanes -> consecutive points: 16*GROUP-byte runs
                const uint32_t idx = q * 64u + lane;
                const uint32_t src = idx / GROUP, pt = idx % GROUP;
                const uint64_t p = span0 + (uint64_t)src * CH + st + pt;
                if (p < a.total) g_out[p] = stage[wave][src * PITCH + pt];  // (nontemporal stores: 6 % slower)
            }
        }
        wave_sync();
    }
    if (bad != ~0ull) atomicMin(g_bad, (unsigned long long)bad);
    if ((KT_CGR_DEBUG & 2) && mx == 1234.5) g_out[s] = make_double2(mx, my);
}

}  // namespace

using namespace ktl;

extern "C" int kt_cgr_points(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                             double vecsize, double *xy, uint64_t *bad_pos, int mem) {
    if (!ctx) return kt::fail(KT_ERR_ARG, "kt_cgr_points: null ctx");
    if (mem != KT_MEM_HOST && mem != KT_MEM_DEVICE) return kt::fail(KT_ERR_ARG, "kt_cgr_points: bad mem");
    if (!(vecsize >= 0.0)) return kt::fail(KT_ERR_ARG, "kt_cgr_points: vecsize must be >= 0");
    if (int rc = ctx->use()) return rc;
    uint64_t total = 0;
    if (n_reads) {
        if (!offsets) return kt::fail(KT_ERR_ARG, "kt_cgr_points: null offsets");
        if (int rc = total_bases_of(ctx, offsets, n_reads, mem, &total)) return rc;
    }
    // device scalar for the first bad index
    if (int rc = ctx->s_aux2.reserve(64)) return rc;
    unsigned long long *d_bad = (unsigned long long *)ctx->s_aux2.p;
    KT_HIP(hipMemsetAsync(d_bad, 0xFF, 8, ctx->stream));
    if (total) {
        if (!bases || !xy) return kt::fail(KT_ERR_ARG, "kt_cgr_points: null buffer");
        const uint8_t *d_bases = bases;
        const uint64_t *d_offsets = offsets;
        double *d_out = xy;
        if (mem == KT_MEM_HOST) {
            if (int rc = stage_batch(ctx, bases, offsets, n_reads, &d_bases, &d_offsets)) return rc;
            if (int rc = ctx->s_out.reserve(total * 16)) return rc;
            d_out = (double *)ctx->s_out.p;
        }
        ktseg::SegArgs sa;
        if (int rc = make_seg_args(ctx, d_bases, d_offsets, n_reads, total, 1, &sa)) return rc;
        CgrShape a{n_reads, total, vecsize};
        const uint64_t n_span = (total + SPAN - 1) / SPAN;
        const uint32_t grid = (uint32_t)((n_span + WAVES - 1) / WAVES);
        hipLaunchKernelGGL(cgr_kernel, dim3(grid), dim3(BLOCK), 0, ctx->stream, d_bases, d_offsets, sa.seg_first, a,
                           (double2 *)d_out, d_bad);
        KT_HIP(hipGetLastError());
        if (mem == KT_MEM_HOST) KT_HIP(hipMemcpyAsync(xy, d_out, total * 16, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (mem == KT_MEM_DEVICE) {
        if (bad_pos) KT_HIP(hipMemcpyAsync(bad_pos, d_bad, 8, hipMemcpyDeviceToDevice, ctx->stream));
        return KT_OK;
    }
    uint64_t bad = ~0ull;
    KT_HIP(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    KT_HIP(hipStreamSynchronize(ctx->stream));
    if (bad_pos) *bad_pos = bad;
    if (bad != ~0ull) return kt::fail(KT_ERR_BADNT, "Bad nucleotide, unable to proceed");  // cgr.rs:140
    return KT_OK;
}
