// kt_min.hip - window minimisers of every read (the `min` subcommand, python MinimiserGenerator).
//
// Replaces MinimiserGenerator::next (reference kmer/src/minimiser.rs:61-175), a stateful iterator
// that yields (minimiser, window start, window end) whenever the smallest canonical m-mer of the
// sliding w-base window changes.  Restated position-parallel (the tests check it against a
// statement-for-statement port of the iterator on random inputs - tests/test_gpu_parity.py):
//   run      maximal stretch of unambiguous bases inside one read; run_len(p) = bases of the
//            run up to and including p
//   val(p)   canonical m-mer ending at p (run_len(p) >= m)
//   act(p)   min(val(p - W + 1 .. p)), W = w - m + 1: the active minimiser once the window is
//            full, i.e. run_len(p) >= w
//   events   E1  run_len(p) > w and act(p) != act(p-1)        -> (act(p-1), ws, p)
//            E2  base p is ambiguous and run_len(p-1) >= w     -> (act(p-1), ws, p)
//            E3  p is the last base of its read, run_len(p) >= m, and no E1 at p
//                                                              -> (act(p) or u64::MAX if the
//                                                                  window never filled, ws, n)
//            ws = (position of the previous E1 of the same run) - w + 1, else the run's start
//            (quirks kept: an E1 on the last base swallows the read's final window; a read whose
//            last run is shorter than w reports u64::MAX)
// w = 0 ("one minimiser per sequence", misc/src/minimisers.rs:44-48) makes w the read's own length;
// it has its own one-thread-per-read kernel.
//
// General w: the batch is cut into tiles of 7168 positions + a 1024-position halo in front (W <= 1024; tiles of
// 4096 + a 4096-position halo for W <= 4096), 16 consecutive positions per thread (one 16-byte load, SWAR-encoded).  Run starts
// are a max-scan ("latest break before p") carried across tiles by a small prefix pass over
// 1024-position granules; the sliding minimum is log2(W) doubling steps over a padded LDS array
// of the tile's m-mers.  Output is dense and in read order: with a known capacity one pass does it
// all - tiles take tickets, publish their event counts and chain their output offsets by
// decoupled look-back - and a capacity-0 call is the count-only pass.  A last small kernel
// resolves window starts and read-local coordinates.  HBM traffic is a few bytes per base; the
// work is LDS/VALU bound.  No MFMA.
#include "kt_internal.hpp"
#include "kt_launch.hpp"

#ifndef KT_MIN_DEBUG
#define KT_MIN_DEBUG 0  // timing experiments only: 1 = no event stores, 2 = no per-read offsets, 4 = no look-back
#endif

namespace {

#ifndef KT_MIN_BLOCK
#define KT_MIN_BLOCK 512
#endif
constexpr int BLOCK = KT_MIN_BLOCK;
constexpr uint32_t GRAN = 1024;               // carry granule (positions)
constexpr uint32_t PER = 16;                  // consecutive positions per thread
constexpr uint32_t RANGE = BLOCK * PER;       // positions a workgroup looks at: one halo granule + its tile
constexpr uint32_t MAX_HG = 4;                // widest halo: windows of up to 4096 m-mers
constexpr uint64_t NONE = ~0ull;
static_assert(PER * BLOCK == RANGE, "tiling");

struct MinArgs {
    const uint8_t *bases;
    const uint64_t *offsets;
    const uint64_t *gfirst;   // [n_gran + 1] first read r with offsets[r] >= g * GRAN (n_reads if none)
    const uint64_t *carry;    // [n_gran] latest break (run start candidate) + 1 before granule g, 0 = none
    uint64_t n_reads, total, n_gran;
    uint32_t w, m, W;
};

// ---- granule index / carries ---------------------------------------------------------------------
__global__ void gran_index_kernel(const uint64_t *__restrict__ offsets, uint64_t n_reads, uint64_t *__restrict__ gfirst,
                                  uint64_t n_gran) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_reads) return;
    const uint64_t cur = offsets[r];
    uint64_t g_lo = (r == 0) ? 0 : offsets[r - 1] / GRAN + 1;
    uint64_t g_hi = cur / GRAN;
    if (r == n_reads) g_hi = n_gran;
    if (g_hi > n_gran) g_hi = n_gran;
    for (uint64_t g = g_lo; g <= g_hi; g++) gfirst[g] = r;
}

// lastbreak[g] = 1 + the latest run-start candidate inside granule g (a read start q gives q, an ambiguous
// base q gives q + 1), 0 if the granule has none.  One wave per granule: a lane checks 16 bases (SWAR).
__global__ __launch_bounds__(BLOCK) void gran_break_kernel(const uint8_t *__restrict__ bases, uint64_t total,
                                                           const uint64_t *__restrict__ offsets,
                                                           const uint64_t *__restrict__ gfirst, uint64_t n_reads,
                                                           uint64_t n_gran, uint64_t *__restrict__ lastbreak) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t g = (uint64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (g >= n_gran) return;
    const uint64_t p0 = g * GRAN, p = p0 + 16ull * lane;
    unsigned long long mine = 0;
    if (p < total) {
        uint32_t d[4];
        if (p + 16 <= total) {
            __builtin_memcpy(d, bases + p, 16);
        } else {
            unsigned char raw[16];
            for (int j = 0; j < 16; j++) raw[j] = p + j < total ? bases[p + j] : (unsigned char)'A';
            __builtin_memcpy(d, raw, 16);
        }
        uint32_t inv = 0, any_raw = 0;  // inv: bit 15 - j = byte j is not a nucleotide
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t c8, i4, rw;
            ktd::swar4(d[q], c8, i4, rw);
            inv = (inv << 4) | i4;
            any_raw |= rw;
        }
        if (any_raw) {  // raw codes 0..3 are nucleotides too: per-byte check
            inv = 0;
            for (int j = 0; j < 16; j++) inv |= (ktd::nt4((d[j >> 2] >> (8 * (j & 3))) & 0xFFu) >> 2) << (15 - j);
        }
        if (inv) mine = p + (15u - (uint32_t)__builtin_ctz(inv)) + 2;  // last ambiguous byte q: candidate q + 1, stored + 1
    }
    for (uint64_t r = gfirst[g] + lane; r < n_reads; r += 64) {
        const uint64_t o = offsets[r];
        if (o >= p0 + GRAN) break;
        if (o + 1 > mine) mine = o + 1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(mine, off, 64);
        mine = o > mine ? o : mine;
    }
    if (lane == 0) lastbreak[g] = mine;
}

// ---- device-wide exclusive scans over small arrays (sum or max), three launches ------------------------
template <bool MAX>
__device__ __forceinline__ uint64_t comb(uint64_t a, uint64_t b) {
    return MAX ? (a > b ? a : b) : a + b;
}

// inclusive scan of one value per thread across the 1024-thread workgroup; returns the exclusive prefix
template <bool MAX>
__device__ uint64_t block_scan1024(uint64_t v, uint64_t *total, uint64_t *tmp /*16*/) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc = comb<MAX>(inc, o);
    }
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (uint32_t w = 0; w < blockDim.x / 64; w++) {
        if (w < wave) base = comb<MAX>(base, tmp[w]);
        tot = comb<MAX>(tot, tmp[w]);
    }
    __syncthreads();
    if (total) *total = tot;
    const uint64_t prev = __shfl_up(inc, 1, 64);
    return comb<MAX>(base, lane ? prev : 0);
}

template <bool MAX>
__global__ __launch_bounds__(1024) void scan_reduce_kernel(const uint64_t *__restrict__ in, uint64_t n,
                                                           uint64_t *__restrict__ partial) {
    __shared__ uint64_t tmp[16];
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    uint64_t tot;
    block_scan1024<MAX>(i < n ? in[i] : 0, &tot, tmp);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of partial[0..np) in place, grand total to *total
template <bool MAX>
__global__ __launch_bounds__(1024) void scan_partials_kernel(uint64_t *__restrict__ partial, uint64_t np,
                                                             uint64_t *__restrict__ total) {
    __shared__ uint64_t tmp[16];
    const uint64_t per = (np + 1023) / 1024;
    const uint64_t lo = (uint64_t)threadIdx.x * per, hi = lo + per < np ? lo + per : np;
    uint64_t acc = 0;
    for (uint64_t i = lo; i < hi; i++) acc = comb<MAX>(acc, partial[i]);
    uint64_t tot;
    uint64_t run = block_scan1024<MAX>(acc, &tot, tmp);
    for (uint64_t i = lo; i < hi; i++) {
        const uint64_t v = partial[i];
        partial[i] = run;
        run = comb<MAX>(run, v);
    }
    if (threadIdx.x == 0 && total) *total = tot;
}

template <bool MAX>
__global__ __launch_bounds__(1024) void scan_apply_kernel(const uint64_t *__restrict__ in, uint64_t n,
                                                          const uint64_t *__restrict__ partial,
                                                          uint64_t *__restrict__ out) {
    __shared__ uint64_t tmp[16];
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    const uint64_t ex = block_scan1024<MAX>(i < n ? in[i] : 0, nullptr, tmp);
    if (i < n) out[i] = comb<MAX>(partial[blockIdx.x], ex);
}

// out[i] = exclusive scan of in[0..n); *total (device, may be null) = grand total.  partial: >= n/1024 + 1 entries
template <bool MAX>
int device_excl_scan(kt_ctx *ctx, const uint64_t *in, uint64_t n, uint64_t *out, uint64_t *partial, uint64_t *total) {
    if (n == 0) {
        if (total) KT_HIP(hipMemsetAsync(total, 0, 8, ctx->stream));
        return KT_OK;
    }
    const uint64_t nb = (n + 1023) / 1024;
    hipLaunchKernelGGL(scan_reduce_kernel<MAX>, dim3((uint32_t)nb), dim3(1024), 0, ctx->stream, in, n, partial);
    hipLaunchKernelGGL(scan_partials_kernel<MAX>, dim3(1), dim3(1024), 0, ctx->stream, partial, nb, total);
    hipLaunchKernelGGL(scan_apply_kernel<MAX>, dim3((uint32_t)nb), dim3(1024), 0, ctx->stream, in, n, partial, out);
    KT_HIP(hipGetLastError());
    return KT_OK;
}

// ---- the tile kernel ------------------------------------------------------------------------------------
struct Event {
    uint64_t val, pos, run, rs;  // minimiser, global position of the event, global start of its run and of its read
};

// LDS index of local position li: one pad entry per 16, because every thread works on 16 consecutive
// positions - unpadded, the 64 lanes of a wave would all hit the same bank (measured: 98 us per tile)
__device__ __forceinline__ uint32_t ph(uint32_t li) { return li + (li >> 4); }

template <class V>
struct TileShared {
    V a[RANGE + RANGE / 16];     // m-mers -> doubling -> active minimiser (padded, see ph())
    uint32_t packed[BLOCK + 2];  // 16 bases per word (2-bit codes, first base on top); [0], [1] = the 32 bases before the range
    uint32_t start_bits[RANGE / 32 + 1];  // bit per position of the range (+1): a read starts here
    uint16_t rank[RANGE];        // EMIT: events of the tile in front of each position
    int32_t scan_tmp[2][8];
    uint32_t cnt_tmp[8];
};

// exclusive max-scan of one value per thread over the 256-thread workgroup (identity: lowest)
__device__ __forceinline__ int32_t block_max_excl(int32_t v, int32_t lowest, int32_t *tmp) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    int32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off && o > inc) inc = o;
    }
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    int32_t base = lowest;
    for (uint32_t w = 0; w < wave; w++) base = base > tmp[w] ? base : tmp[w];
    const int32_t prev = __shfl_up(inc, 1, 64);
    const int32_t ex = lane ? prev : lowest;
    return base > ex ? base : ex;
}

__device__ __forceinline__ uint32_t block_sum_excl(uint32_t v, uint32_t *tmp, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t w = 0; w < BLOCK / 64; w++) {
        if (w < wave) base += tmp[w];
        tot += tmp[w];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// V = uint32_t for m <= 16 (an m-mer fits 32 bits; all-ones is never canonical, so it marks "none"),
// uint64_t otherwise.  Everything inside the tile is in 32-bit local coordinates (index into the range);
// "break" values are run / read start candidates as local indices, FAR when they lie in front of the range
// (their exact global values then come from the carries).
// EMIT = false: count pass, tile_count[tile] = events of the tile.
// EMIT = true:  single pass.  Tiles take their number from a ticket counter (so a tile only ever waits for
// tiles that are already running), publish their event count, and find their output offset by looking back
// over the predecessors' published counts / prefixes (decoupled look-back, one wave, 64 tiles per step);
// no separate count pass or scan is needed when the caller's capacity is known to suffice.
struct Chain {
    unsigned long long *status;  // [n_tiles] (count or inclusive prefix) | flag in the top two bits; zeroed
    unsigned long long *ticket;  // [1] zeroed
    uint64_t *total;             // [1] receives the number of events
    uint64_t n_tiles, capacity;
};
constexpr unsigned long long ST_AGG = 1ull << 62, ST_PREFIX = 2ull << 62, ST_MASK = 3ull << 62;

// HG = granules of halo in front of the tile: windows of up to HG * 1024 m-mers (1: tiles of 7168 positions;
// 4: tiles of 4096, for the rare wide windows)
// MODE (windows of more than MAX_HG * 1024 m-mers - the two-level sliding minimum, see kt_minimisers): 1 = the tile's
// sliding minimum over a.W (= 4096) m-mers goes to act[] and nothing else happens; 2 = the active minimiser of every
// position comes from act[] (strided_min_kernel has made it) and the tile does everything but the sliding minimum
template <class V, bool EMIT, int HG, int MODE = 0>
__global__ __launch_bounds__(BLOCK) void min_tile_kernel(MinArgs a, uint64_t *__restrict__ tile_count, Chain chain,
                                                         Event *__restrict__ ev, uint8_t *__restrict__ ev_type,
                                                         uint64_t *__restrict__ ev_offsets, V *__restrict__ act) {
    __shared__ TileShared<V> sm;
    __shared__ uint64_t sh_u64;
    constexpr V VNONE = (V)~(V)0;
    constexpr int32_t FAR = -(1 << 30);
    const uint32_t tid = threadIdx.x;
    uint64_t tile = blockIdx.x;
    if (EMIT) {
        if (tid == 0) sh_u64 = atomicAdd(chain.ticket, 1ull);
        __syncthreads();
        tile = sh_u64;
    }
    constexpr uint32_t HALO = HG * GRAN, TILE = RANGE - HALO;
    static_assert(HALO <= TILE, "only tile 0 may start in front of the batch");
    const uint64_t t0 = tile * TILE;           // first position owned
    const int64_t range0 = (int64_t)t0 - (int64_t)HALO;        // position of local index 0 (negative for tile 0)
    const int32_t m = (int32_t)a.m, w = (int32_t)a.w;
    const uint32_t W = a.W;
    // local indices [lo_in, hi_in) are real positions (tile 0 has no halo, the last tile may be short)
    const int32_t lo_in = range0 < 0 ? (int32_t)HALO : 0;
    const int32_t hi_in = a.total - t0 >= TILE ? (int32_t)RANGE : (int32_t)(a.total - t0) + (int32_t)HALO;

    // ---- this thread's 16 bases: one 16-byte load, SWAR-encoded (codes P, ambiguity flags inv) ----
    const int32_t l0 = (int32_t)(tid * PER);  // first local index of this thread
    // P: base j at bits 31-2j..30-2j; inv: bit 15-j set = base j is not a nucleotide (or outside the batch)
    auto encode16 = [&](int64_t p, uint32_t &P, uint32_t &inv) {
        uint32_t d[4];
        if (p >= 0 && (uint64_t)p + 16 <= a.total) {
            __builtin_memcpy(d, a.bases + p, 16);
        } else {
            unsigned char raw[16];
            for (int j = 0; j < 16; j++) raw[j] = (p + j >= 0 && (uint64_t)(p + j) < a.total) ? a.bases[p + j] : (unsigned char)'N';
            __builtin_memcpy(d, raw, 16);
        }
        uint32_t any_raw = 0;
        P = 0;
        inv = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t c8, i4, rw;
            ktd::swar4(d[q], c8, i4, rw);
            P = (P << 8) | c8;
            inv = (inv << 4) | i4;
            any_raw |= rw;
        }
        if (any_raw) {  // raw codes 0..3 in the input: per-byte table
            P = 0;
            inv = 0;
            for (int j = 0; j < 16; j++) {
                const uint32_t e = ktd::nt4((d[j >> 2] >> (8 * (j & 3))) & 0xFFu);
                P |= (e & 3u) << (30 - 2 * j);
                inv |= (e >> 2) << (15 - j);
            }
        }
    };
    uint32_t P, inv16;
    encode16(range0 + (int64_t)l0, P, inv16);
    sm.packed[tid + 2] = P;
    if (tid < 2) {
        uint32_t Pc, ic;
        encode16(range0 - 32 + 16 * (int64_t)tid, Pc, ic);
        sm.packed[tid] = Pc;
    }
    for (uint32_t i = tid; i < RANGE / 32 + 1; i += BLOCK) sm.start_bits[i] = 0;
    __syncthreads();
    {
        const uint64_t lo_pos = range0 < 0 ? 0 : (uint64_t)range0;
        const uint64_t end = t0 + TILE + 1;  // one past the tile: "is p + 1 a read start"
        for (uint64_t r = a.gfirst[lo_pos / GRAN] + tid; r < a.n_reads; r += BLOCK) {
            const uint64_t o = a.offsets[r];
            if (o >= end) break;
            const uint32_t rel = (uint32_t)((int64_t)o - range0);
            atomicOr(&sm.start_bits[rel >> 5], 1u << (rel & 31u));
        }
    }
    __syncthreads();

    // code / ambiguity of the thread's position j (compile-time j: bit-field extracts)
#define KT_CODE(j) ((P >> (30 - 2 * (j))) & 3u)
#define KT_AMBIG(j) ((inv16 >> (15 - (j))) & 1u)
    // bit j: a read starts at position l0 + j (bit 16: at the position after this thread's last)
    const uint32_t sb = (uint32_t)((((uint64_t)sm.start_bits[(l0 >> 5) + 1] << 32) | sm.start_bits[l0 >> 5]) >> (l0 & 31)) &
                        0x1FFFFu;
    uint32_t inside = 0;
#pragma unroll
    for (int32_t j = 0; j < (int32_t)PER; j++) inside |= (l0 + j >= lo_in && l0 + j < hi_in ? 1u : 0u) << j;

    // ---- run / read starts: thread-local pass, workgroup max-scans, carries from in front of the range ----
    int32_t loc_run = FAR, loc_read = FAR;  // latest run-start / read-start candidate among this thread's positions
#pragma unroll
    for (int32_t j = 0; j < (int32_t)PER; j++) {
        if (!((inside >> j) & 1u)) continue;
        if ((sb >> j) & 1u) loc_run = loc_read = l0 + j;
        if (KT_AMBIG(j)) loc_run = l0 + j + 1;
    }
    int32_t run0 = block_max_excl(loc_run, FAR, sm.scan_tmp[0]);          // as of the position in front of l0
    const int32_t read0 = block_max_excl(loc_read, FAR, sm.scan_tmp[1]);
    // starts in front of the range come from the carries.  A run start shortly before the range keeps its
    // (negative) local index: with w > 1024 an owned position can still be inside that run's first w bases,
    // and the run lengths must be exact there; anything further away collapses to FAR.
    const uint64_t carry_run = range0 > 0 ? a.carry[(uint64_t)range0 / GRAN] - 1 : 0;
    if (range0 > 0) {
        const int64_t rel = (int64_t)carry_run - range0;
        const int32_t carry_local = rel < (int64_t)FAR ? FAR : (int32_t)rel;
        run0 = run0 > carry_local ? run0 : carry_local;
    }
    uint64_t carry_read = 0;
    if (range0 > 0) {  // the read that holds position range0 - 1
        const uint64_t g = (uint64_t)range0 / GRAN;
        uint64_t r = a.gfirst[g];  // first read starting at or after range0
        carry_read = a.offsets[r - 1];  // r >= 1: read 0 starts at 0 < range0
    }
    auto global_of = [&](int32_t loc, uint64_t far_value) {
        return loc == FAR ? far_value : (uint64_t)(range0 + (int64_t)loc);
    };

    if constexpr (MODE == 2) {
        // the active minimisers of the owned positions and of the one in front of them, as the passes before left them
#pragma unroll
        for (int32_t j = 0; j < (int32_t)PER; j++) {
            const int32_t li = l0 + j;
            sm.a[ph((uint32_t)li)] = ((inside >> j) & 1u) && li + 1 >= (int32_t)HALO ? act[range0 + (int64_t)li] : VNONE;
        }
        __syncthreads();
    }
    // ---- canonical m-mers of this thread's positions -> sm.a ----
    if constexpr (MODE != 2) {
        V f = 0, r = 0;
        const V mask = (V)(((uint64_t)1 << (2 * m)) - 1ull);
        const uint32_t rsh = 2 * (uint32_t)(m - 1);
        // the m - 1 codes in front of the first position = the low bits of the two packed words before this
        // thread's (ambiguous ones are harmless: run_len gates the use); their reverse complement seeds r
        if (m > 1) {
            const uint64_t ctx = ((uint64_t)sm.packed[tid] << 32) | sm.packed[tid + 1];
            const uint64_t fm = ctx & ((1ull << (2 * (m - 1))) - 1ull);
            f = (V)fm;
            r = (V)(ktd::rev_comp(fm, m - 1) << 2);
        }
        int32_t b = run0;
#pragma unroll
        for (int32_t j = 0; j < (int32_t)PER; j++) {
            const uint32_t c = KT_CODE(j);
            f = (V)(((f << 2) | c) & mask);
            r = (V)((r >> 2) | ((V)(3u - c) << rsh));
            V v = VNONE;
            if ((inside >> j) & 1u) {
                if ((sb >> j) & 1u) b = l0 + j;
                if (KT_AMBIG(j)) b = l0 + j + 1;
                if (l0 + j + 1 - b >= m) v = f < r ? f : r;  // run_len = position + 1 - run start
            }
            sm.a[ph((uint32_t)(l0 + j))] = v;
        }
    }
    if constexpr (MODE != 2) __syncthreads();

    // ---- sliding minimum over W m-mers: doubling, then two overlapping power-of-two windows ----
    if constexpr (MODE != 2) {
        // only windows ending at local index >= HALO - 1 are ever read; the values they are built from reach
        // back less than 2 W positions, so the threads further in front (most of the halo wave) just keep step
        const bool needed = (uint32_t)l0 + PER + 2 * W + 2 > HALO;
        auto combine = [&](uint32_t dist) {
            V x[PER];
            if (needed) {
#pragma unroll
                for (uint32_t j = 0; j < PER; j++) {
                    const uint32_t li = (uint32_t)l0 + j;
                    const V u = sm.a[ph(li)];
                    const V o = li >= dist ? sm.a[ph(li - dist)] : VNONE;
                    x[j] = u < o ? u : o;
                }
            }
            __syncthreads();
            if (needed) {
#pragma unroll
                for (uint32_t j = 0; j < PER; j++) sm.a[ph((uint32_t)l0 + j)] = x[j];
            }
            __syncthreads();
        };
        uint32_t span = 1;  // sm.a[i] = min of the `span` m-mers ending at i
        while (span * 2 <= W) {
            combine(span);
            span *= 2;
        }
        if (span < W) combine(W - span);
    }

    if constexpr (MODE == 1) {
#pragma unroll
        for (int32_t j = 0; j < (int32_t)PER; j++) {
            const int32_t li = l0 + j;
            if (((inside >> j) & 1u) && li >= (int32_t)HALO) act[range0 + (int64_t)li] = sm.a[ph((uint32_t)li)];
        }
        return;
    }

    // ---- events of the positions this workgroup owns (local index >= HALO) ----
    // detection only: which of the thread's positions emit (ev_bits), of which kind (2 bits each), and whether an
    // E3 reports "no full window" (none_bits).  The records are built afterwards, one per set bit.
    uint32_t ev_bits = 0, kinds = 0, none_bits = 0, inv_bits = 0;
    {
        V act_prev = l0 ? sm.a[ph((uint32_t)l0 - 1)] : VNONE;
        int32_t b = run0;
#pragma unroll
        for (int32_t j = 0; j < (int32_t)PER; j++) {
            const V act = sm.a[ph((uint32_t)(l0 + j))];
            if ((inside >> j) & 1u) {
                const int32_t li = l0 + j;
                const int32_t b_prev = b;  // run bookkeeping as of the previous position
                const bool st = (sb >> j) & 1u;
                if (st) b = li;
                if (KT_AMBIG(j)) {
                    b = li + 1;
                    inv_bits |= 1u << j;
                }
                if (li >= (int32_t)HALO) {  // halo positions only carry state
                    const int32_t run_len = li + 1 - b;
                    // run length of the previous position (0 at a read start: that base belongs to another read)
                    const int32_t prev_len = st ? 0 : li - b_prev;
                    const bool last = (li + 1 == hi_in && t0 + TILE >= a.total) || ((sb >> (j + 1)) & 1u);
                    uint32_t kind = 0;
                    if (KT_AMBIG(j)) {
                        if (prev_len >= w) kind = 2;                    // E2: an ambiguous base closes a full window
                    } else if (run_len > w && act != act_prev) {
                        kind = 1;                                       // E1: the active minimiser changed
                    } else if (last && run_len >= m) {
                        kind = 3;                                       // E3: the read's last window
                        if (run_len < w) none_bits |= 1u << j;
                    }
                    if (kind) {
                        ev_bits |= 1u << j;
                        kinds |= kind << (2 * j);
                    }
                }
            }
            act_prev = act;
        }
    }
    const uint32_t n_ev = (uint32_t)__builtin_popcount(ev_bits);
    uint32_t tile_total;
    const uint32_t rank0 = block_sum_excl(n_ev, sm.cnt_tmp, &tile_total);
    if (!EMIT) {
        if (tid == 0) tile_count[tile] = tile_total;
        return;
    }
    // ---- output offset of this tile: publish the count, look back over the predecessors ----
    if (tid < 64) {
        if (tid == 0)
            // the status word carries its own payload, so relaxed device-scope atomics are enough (acquire loads
            // in the polling loop invalidated the L1 on every probe: 50 ms instead of 15)
            __hip_atomic_store(&chain.status[tile], (unsigned long long)tile_total | (tile ? ST_AGG : ST_PREFIX),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t sum = 0;
        int64_t j0 = (KT_MIN_DEBUG & 4) ? -1 : (int64_t)tile - 1;  // lane l looks at tile j0 - l
        while (j0 >= 0) {
            const int64_t j = j0 - (int64_t)tid;
            unsigned long long st = ST_PREFIX;  // tiles in front of tile 0: an empty prefix
            if (j >= 0) {
                do {
                    st = __hip_atomic_load(&chain.status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while ((st & ST_MASK) == 0);
            }
            const uint64_t is_prefix = __ballot((st & ST_MASK) == ST_PREFIX);
            const uint32_t first = is_prefix ? (uint32_t)__builtin_ctzll(is_prefix) : 64u;  // nearest tile with a prefix
            uint64_t part = tid <= first ? (uint64_t)(st & ~ST_MASK) : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
            sum += __shfl(part, 0, 64);
            if (is_prefix) break;
            j0 -= 64;
        }
        if (tid == 0) {
            if (tile)
                __hip_atomic_store(&chain.status[tile], (unsigned long long)(sum + tile_total) | ST_PREFIX, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            if (tile + 1 == chain.n_tiles) *chain.total = sum + tile_total;
            sh_u64 = sum;
        }
    }
    __syncthreads();
    const uint64_t base = sh_u64;
    {
        // one record per set bit of ev_bits.  Everything a record needs is a function of the bit masks: the run
        // start is the latest break (read start -> its position, ambiguous base -> the position after it) at or,
        // for an E2, before the event's position; the read start likewise; the minimiser sits in sm.a.
        const uint32_t starts_in = sb & inside & 0xFFFFu;
        uint32_t bits = ev_bits, rk = rank0;
        while (bits) {
            const uint32_t j = (uint32_t)__builtin_ctz(bits);
            bits &= bits - 1;
            const uint32_t kind = (kinds >> (2 * j)) & 3u;
            const uint32_t li = (uint32_t)l0 + j;
            V val = sm.a[ph(kind == 3 ? li : li - 1)];
            if ((none_bits >> j) & 1u) val = VNONE;
            const uint32_t upto = (2u << j) - 1u;                     // positions <= j
            const uint32_t before = kind == 2 ? (upto >> 1) : upto;   // an E2 closes the run that ended before it
            const uint32_t sm_run = starts_in & before, im_run = inv_bits & before, sm_read = starts_in & upto;
            int32_t run = run0;
            if (sm_run) run = max(run, l0 + 31 - (int32_t)__builtin_clz(sm_run));
            if (im_run) run = max(run, l0 + 32 - (int32_t)__builtin_clz(im_run));
            const int32_t rd = sm_read ? l0 + 31 - (int32_t)__builtin_clz(sm_read) : read0;
            if (base + rk < chain.capacity && !(KT_MIN_DEBUG & 1)) {
                ev[base + rk] = Event{val == VNONE ? NONE : (uint64_t)val, (uint64_t)(range0 + (int64_t)li),
                                      global_of(run, carry_run), global_of(rd, carry_read)};
                ev_type[base + rk] = (uint8_t)kind;
            }
            rk++;
        }
        // events in front of every position (for the reads that start inside the tile)
        uint32_t before_pos = rank0;
#pragma unroll
        for (uint32_t j = 0; j < PER; j++) {
            sm.rank[l0 + j] = (uint16_t)before_pos;
            before_pos += (ev_bits >> j) & 1u;
        }
    }
    __syncthreads();
    if (KT_MIN_DEBUG & 2) return;
    for (uint64_t r = a.gfirst[t0 / GRAN] + tid; r < a.n_reads; r += BLOCK) {
        const uint64_t o = a.offsets[r];
        if (o >= t0 + TILE || o >= a.total) break;
        ev_offsets[r] = base + sm.rank[(uint32_t)((int64_t)o - range0)];
    }
}

// out[p] = min(in[p], in2[p - d]) (nothing in front of the batch): one step of the second level of the sliding minimum
template <class V>
__global__ __launch_bounds__(256) void strided_min_kernel(const V *__restrict__ in, const V *__restrict__ in2, V *__restrict__ out,
                                                          uint64_t n, uint64_t d) {
    for (uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (uint64_t)gridDim.x * 256) {
        const V u = in[p];
        const V o = p >= d ? in2[p - d] : (V)~(V)0;
        out[p] = u < o ? u : o;
    }
}

// reads that start at the very end of the batch (empty, after the last base) were not seen by any tile
__global__ void min_tail_kernel(uint64_t *__restrict__ ev_offsets, uint64_t n_reads, const uint64_t *__restrict__ total) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_reads) return;
    if (r == n_reads || ev_offsets[r] == NONE) ev_offsets[r] = *total;
}

// one thread per event: window start (the previous change of the same run) and read-local coordinates
__global__ void min_finalize_kernel(const Event *__restrict__ ev, const uint8_t *__restrict__ ev_type, uint64_t n_ev,
                                    uint32_t w, uint64_t capacity, uint64_t *__restrict__ kmers,
                                    uint64_t *__restrict__ starts, uint64_t *__restrict__ ends) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_ev || j >= capacity) return;
    const Event e = ev[j];
    uint64_t ws = e.run;
    if (j > 0 && ev_type[j - 1] == 1) {
        const Event q = ev[j - 1];
        if (q.run == e.run && q.rs == e.rs) ws = q.pos + 1 - w;
    }
    kmers[j] = e.val;
    starts[j] = ws - e.rs;
    ends[j] = (ev_type[j] == 3 ? e.pos + 1 : e.pos) - e.rs;
}

// ---- w = 0: one window per read ------------------------------------------------------------------------------
// count[r] = 1 if the read reports a minimiser: its last run holds >= m bases (and the read >= m bases)
template <bool EMIT>
__global__ __launch_bounds__(BLOCK) void min_whole_kernel(const uint8_t *__restrict__ bases,
                                                          const uint64_t *__restrict__ offsets, uint64_t n_reads,
                                                          uint32_t m, uint64_t *__restrict__ count,
                                                          const uint64_t *__restrict__ ev_offsets, uint64_t capacity,
                                                          uint64_t *__restrict__ kmers, uint64_t *__restrict__ starts,
                                                          uint64_t *__restrict__ ends) {
    const uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t s = offsets[r], e = offsets[r + 1], n = e - s;
    const uint64_t mask = (1ull << (2 * m)) - 1ull;
    const uint32_t rsh = 2 * (m - 1);
    uint64_t f = 0, rv = 0, run = 0, best = NONE, run_start = 0;
    bool clean = true;  // no ambiguous base so far: the single window (w = n) can still fill
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t c = ktd::nt4(bases[s + i]);
        if (c > 3) {
            run = 0;
            run_start = i + 1;
            clean = false;
            continue;
        }
        f = ((f << 2) | c) & mask;
        rv = (rv >> 2) | ((uint64_t)(3u - c) << rsh);
        run++;
        if (run >= m) {
            const uint64_t v = f < rv ? f : rv;
            best = v < best ? v : best;
        }
    }
    const bool has = n >= m && run >= m;  // minimiser.rs:103-106, :168-171
    if (!EMIT) {
        count[r] = has ? 1 : 0;
        return;
    }
    if (has) {
        const uint64_t j = ev_offsets[r];
        if (j < capacity) {
            kmers[j] = clean ? best : NONE;  // the window (w = n) only fills if the whole read is one run
            starts[j] = run_start;
            ends[j] = n;
        }
    }
}

// ---- windows of more than 4096 m-mers: one thread per read ---------------------------------------------------------
// The tile kernel's halo stops at 4096 positions; wider windows (whole-contig scale) are rare enough that the
// iterator itself is run, one read per thread: MinimiserGenerator::next (kmer/src/minimiser.rs:61-175) with its ring
// buffer in global memory (a read of n bases never holds more than min(n, W) m-mers: ring r lives at
// ring[offsets[r] + r ...]).  The window's rescans are O(W) but happen once per ~W/2 positions, so a read costs
// O(n); parallelism is across reads only.  count != null: count[r] = triples of read r; else they are written.
__global__ __launch_bounds__(BLOCK) void min_serial_kernel(const uint8_t *__restrict__ bases,
                                                           const uint64_t *__restrict__ offsets, uint64_t n_reads,
                                                           uint64_t wsize, uint32_t m, uint64_t *__restrict__ ring_all,
                                                           uint64_t *__restrict__ count,
                                                           const uint64_t *__restrict__ ev_offsets, uint64_t capacity,
                                                           uint64_t *__restrict__ kmers, uint64_t *__restrict__ starts,
                                                           uint64_t *__restrict__ ends) {
    const uint64_t rd = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (rd >= n_reads) return;
    const uint64_t s0 = offsets[rd], len = offsets[rd + 1] - s0;
    const uint64_t full = wsize - m + 1;                 // m-mers of a full window
    const uint64_t cap = (len < full ? len : full) + 1;  // ring slots this read can ever need
    uint64_t *ring = ring_all + s0 + rd;
    const uint64_t mask = (1ull << (2 * m)) - 1ull;
    const uint32_t rsh = 2 * (m - 1);
    uint64_t f = 0, rv = 0, vl = 0;                      // rolling m-mer and its length
    uint64_t head = 0, blen = 0, buff_pos = 0;           // the window's m-mers, position of the active minimum
    uint64_t active = NONE, wstart = 0;
    uint64_t n_out = 0;
    const uint64_t out0 = count ? 0 : ev_offsets[rd];
    auto emit = [&](uint64_t km, uint64_t ws, uint64_t we) {
        if (!count && out0 + n_out < capacity) {
            kmers[out0 + n_out] = km;
            starts[out0 + n_out] = ws;
            ends[out0 + n_out] = we;
        }
        n_out++;
    };
    auto at = [&](uint64_t j) -> uint64_t & { return ring[(head + j) % cap]; };
    for (uint64_t pos = 0; pos < len; pos++) {
        const uint32_t c = ktd::nt4(bases[s0 + pos]);
        if (c > 3) {                                     // :81-101: an ambiguous base ends the run
            const bool ret = blen == full;
            const uint64_t pm = active, pws = wstart;
            buff_pos = 0;
            active = NONE;
            f = rv = vl = 0;
            wstart = pos + 1;
            blen = 0;
            head = 0;
            if (ret) emit(pm, pws, pos);
            continue;
        }
        f = ((f << 2) | c) & mask;                       // :75-80
        rv = (rv >> 2) | ((uint64_t)(3u - c) << rsh);
        vl++;
        if (vl < m) continue;                            // :103-106
        vl--;
        const uint64_t mv = f < rv ? f : rv;             // :110
        bool emitted = false;
        if (blen == full) {                              // :113
            head = (head + 1) % cap;                     // pop_front
            at(blen - 1) = mv;                           // push_back
            if (buff_pos == 0) {                         // :119-138: the minimum left the window
                uint64_t nm = NONE;
                for (uint64_t j = 0; j < blen; j++) {
                    const uint64_t v = at(j);
                    if (v < nm) {
                        buff_pos = j;
                        nm = v;
                    }
                }
                if (nm != active) {
                    emit(active, wstart, pos);
                    active = nm;
                    wstart = pos - wsize + 1;
                    emitted = true;
                }
            } else if (mv < active) {                    // :139-149
                emit(active, wstart, pos);
                active = mv;
                buff_pos = blen - 1;
                wstart = pos - wsize + 1;
                emitted = true;
            } else {
                buff_pos--;                              // :150-152
            }
        } else {
            at(blen) = mv;                               // :153-156
            blen++;
        }
        if (emitted) continue;                           // (the iterator returned: the checks below are skipped)
        if (active == NONE && blen == full) {            // :158-166: the first full window
            for (uint64_t j = 0; j < blen; j++) {
                const uint64_t v = at(j);
                if (v < active) {
                    buff_pos = j;
                    active = v;
                }
            }
        }
        if (pos == len - 1) emit(active, wstart, len);   // :168-171
    }
    if (count) count[rd] = n_out;
}

__global__ void set_last_kernel(uint64_t *__restrict__ ev_offsets, uint64_t n_reads, const uint64_t *__restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ev_offsets[n_reads] = *total;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- windows of more than 4096 m-mers on the tile path: the sliding minimum in two levels --------------------------------
// H = 4096 (what a tile's halo reaches).  Level 1: A0(p) = min of the H m-mers ending at p - the tile kernel itself (MODE 1),
// written out.  Level 2, with W = J H + R: B_J(p) = min over j < J of A0(p - j H) by doubling on the stride-H sequence
// (log2 J elementwise passes, out[p] = min(in[p], in[p - d])), and the window's first H m-mers are A0(p - (W - H)):
// act(p) = min(B_J(p), A0(p - (W - H))).  Windows that reach across a run's start come out wrong and are never looked at
// (an event needs run_len >= w), exactly as inside a tile.  Three arrays of one m-mer per base, in ctx->s_aux0.
template <class V>
int wide_window_act(kt_ctx *ctx, const MinArgs &a_in, V **act_out) {
    const uint64_t total = a_in.total, H = (uint64_t)MAX_HG * GRAN, W = a_in.W;
    const size_t one = align256(total * sizeof(V));
    if (int rc = ctx->s_aux0.reserve(3 * one + 256)) return rc;
    char *base = (char *)ctx->s_aux0.p;
    V *A0 = (V *)base, *bufs[2] = {(V *)(base + one), (V *)(base + 2 * one)};
    MinArgs a = a_in;
    a.W = (uint32_t)H;
    a.w = (uint32_t)H + a.m - 1;
    const uint64_t n_tiles = (total + (RANGE - H) - 1) / (RANGE - H);
    hipLaunchKernelGGL((min_tile_kernel<V, false, (int)MAX_HG, 1>), dim3((uint32_t)n_tiles), dim3(BLOCK), 0, ctx->stream, a,
                       (uint64_t *)nullptr, Chain{}, (Event *)nullptr, (uint8_t *)nullptr, (uint64_t *)nullptr, A0);
    KT_HIP(hipGetLastError());
    const uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256u * 64u);
    int which = 0;
    V *cur = A0;
    auto step = [&](const V *in2, uint64_t d) {
        hipLaunchKernelGGL(strided_min_kernel<V>, dim3(grid), dim3(256), 0, ctx->stream, (const V *)cur, in2, bufs[which], total, d);
        cur = bufs[which];
        which ^= 1;
    };
    const uint64_t J = W / H;
    uint64_t span = 1;
    while (span * 2 <= J) {
        step(cur, span * H);
        span *= 2;
    }
    if (span < J) step(cur, (J - span) * H);
    step(A0, W - H);
    KT_HIP(hipGetLastError());
    *act_out = cur;
    return KT_OK;
}

}  // namespace

using namespace ktl;

extern "C" int kt_minimisers(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                             uint64_t wsize, int msize, uint64_t *ev_offsets, uint64_t *kmers, uint64_t *starts,
                             uint64_t *ends, uint64_t capacity, uint64_t *n_events, int mem) {
    if (!ctx || !n_events) return kt::fail(KT_ERR_ARG, "kt_minimisers: null");
    *n_events = 0;
    if (mem != KT_MEM_HOST && mem != KT_MEM_DEVICE) return kt::fail(KT_ERR_ARG, "kt_minimisers: bad mem");
    if (msize < 1 || msize > 31) return kt::fail(KT_ERR_ARG, "kt_minimisers: msize must be in 1..31");
    if (wsize != 0 && wsize < (uint64_t)msize)
        return kt::fail(KT_ERR_ARG, "kt_minimisers: wsize must be 0 or >= msize");
    // windows of more than 4096 m-mers: the two-level sliding minimum (wide_window_act) in front of the tile kernel; the
    // iterator itself, one read per thread (min_serial_kernel), is what is left for windows of 2^30 bases and more, and
    // what KT_MIN_SERIAL=1 asks for (the tests run both against the oracle)
    const bool beyond = wsize != 0 && wsize - (uint64_t)msize + 1 > MAX_HG * GRAN;
    const char *ser = getenv("KT_MIN_SERIAL");
    bool wide = beyond && (wsize >= (1ull << 30) || (ser && ser[0] == '1'));
    bool two_level = beyond && !wide;
    if (n_reads == 0) return KT_OK;
    if (!offsets || !ev_offsets) return kt::fail(KT_ERR_ARG, "kt_minimisers: null offsets");
    if (capacity && (!kmers || !starts || !ends)) return kt::fail(KT_ERR_ARG, "kt_minimisers: null output");
    if (int rc = ctx->use()) return rc;
    uint64_t total = 0;
    if (int rc = total_bases_of(ctx, offsets, n_reads, mem, &total)) return rc;
    if (total && !bases) return kt::fail(KT_ERR_ARG, "kt_minimisers: null bases");
    if (two_level && ctx->s_aux0.reserve(3 * align256(total * (msize <= 16 ? 4 : 8)) + 256) != KT_OK) {
        kt::set_error("");  // (no room for the three arrays of the two-level minimum: the iterator needs one)
        two_level = false;
        wide = true;
    }

    const uint8_t *d_bases = bases;
    const uint64_t *d_offsets = offsets;
    uint64_t *d_evoff = ev_offsets, *d_k = kmers, *d_s = starts, *d_e = ends;
    if (mem == KT_MEM_HOST) {
        if (int rc = stage_batch(ctx, bases, offsets, n_reads, &d_bases, &d_offsets)) return rc;
        const size_t o1 = align256((n_reads + 1) * 8), oc = align256(capacity * 8);
        if (int rc = ctx->s_out.reserve(o1 + 3 * oc + 256)) return rc;
        char *p = (char *)ctx->s_out.p;
        d_evoff = (uint64_t *)p;
        d_k = (uint64_t *)(p + o1);
        d_s = (uint64_t *)(p + o1 + oc);
        d_e = (uint64_t *)(p + o1 + 2 * oc);
    }

    // internal buffers (one scratch allocation): granule index/carries, per-tile counts/bases, scan partials
    const uint64_t n_gran = (total + GRAN - 1) / GRAN;
    // halo granules in front of every tile: one for windows of up to 1024 m-mers, four for wider ones
    const int hg = (wsize == 0 || two_level || wsize - (uint64_t)msize + 1 <= GRAN) ? 1 : (int)MAX_HG;
    const uint64_t tile_len = RANGE - (uint64_t)hg * GRAN;
    const uint64_t n_tiles = (total + tile_len - 1) / tile_len;
    const uint64_t n_scan = (n_reads > n_gran ? n_reads : n_gran) + 1;
    size_t off = 0;
    const size_t o_gfirst = off;  off += align256((n_gran + 2) * 8);
    const size_t o_break = off;   off += align256((n_gran + 1) * 8);
    const size_t o_carry = off;   off += align256((n_gran + 1) * 8);
    const size_t o_tcount = off;  off += align256((n_tiles + 1) * 8);
    const size_t o_tbase = off;   off += align256((n_tiles + 1) * 8);
    const size_t o_rcount = off;  off += align256((wsize == 0 || wide ? n_reads + 1 : 1) * 8);
    const size_t o_part = off;    off += align256((n_scan / 1024 + 2) * 8);
    const size_t o_total = off;   off += 256;
    if (int rc = ctx->s_aux1.reserve(off)) return rc;
    char *ib = (char *)ctx->s_aux1.p;
    uint64_t *gfirst = (uint64_t *)(ib + o_gfirst), *lastbreak = (uint64_t *)(ib + o_break);
    uint64_t *carry = (uint64_t *)(ib + o_carry), *tcount = (uint64_t *)(ib + o_tcount);
    uint64_t *tbase = (uint64_t *)(ib + o_tbase), *rcount = (uint64_t *)(ib + o_rcount);
    uint64_t *partial = (uint64_t *)(ib + o_part), *d_total = (uint64_t *)(ib + o_total);

    uint64_t n_ev = 0;
    if (wide) {
        if (int rc = ctx->s_aux2.reserve((total + n_reads + 1) * 8)) return rc;
        uint64_t *ring = (uint64_t *)ctx->s_aux2.p;
        const uint32_t nb = (uint32_t)((n_reads + BLOCK - 1) / BLOCK);
        hipLaunchKernelGGL(min_serial_kernel, dim3(nb), dim3(BLOCK), 0, ctx->stream, d_bases, d_offsets, n_reads, wsize,
                           (uint32_t)msize, ring, rcount, (const uint64_t *)nullptr, (uint64_t)0, (uint64_t *)nullptr,
                           (uint64_t *)nullptr, (uint64_t *)nullptr);
        if (int rc = device_excl_scan<false>(ctx, rcount, n_reads, d_evoff, partial, d_total)) return rc;
        hipLaunchKernelGGL(set_last_kernel, dim3(1), dim3(64), 0, ctx->stream, d_evoff, n_reads, d_total);
        KT_HIP(hipMemcpyAsync(&n_ev, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipStreamSynchronize(ctx->stream));
        if (capacity) {
            hipLaunchKernelGGL(min_serial_kernel, dim3(nb), dim3(BLOCK), 0, ctx->stream, d_bases, d_offsets, n_reads, wsize,
                               (uint32_t)msize, ring, (uint64_t *)nullptr, d_evoff, capacity, d_k, d_s, d_e);
            KT_HIP(hipGetLastError());
        }
    } else if (wsize == 0) {
        const uint32_t nb = (uint32_t)((n_reads + BLOCK - 1) / BLOCK);
        hipLaunchKernelGGL(min_whole_kernel<false>, dim3(nb), dim3(BLOCK), 0, ctx->stream, d_bases, d_offsets, n_reads,
                           (uint32_t)msize, rcount, (const uint64_t *)nullptr, (uint64_t)0, (uint64_t *)nullptr,
                           (uint64_t *)nullptr, (uint64_t *)nullptr);
        if (int rc = device_excl_scan<false>(ctx, rcount, n_reads, d_evoff, partial, d_total)) return rc;
        hipLaunchKernelGGL(set_last_kernel, dim3(1), dim3(64), 0, ctx->stream, d_evoff, n_reads, d_total);
        KT_HIP(hipMemcpyAsync(&n_ev, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipStreamSynchronize(ctx->stream));
        if (capacity) {
            hipLaunchKernelGGL(min_whole_kernel<true>, dim3(nb), dim3(BLOCK), 0, ctx->stream, d_bases, d_offsets, n_reads,
                               (uint32_t)msize, (uint64_t *)nullptr, d_evoff, capacity, d_k, d_s, d_e);
            KT_HIP(hipGetLastError());
        }
    } else {
        KT_HIP(hipMemsetAsync(d_evoff, 0xFF, (n_reads + 1) * 8, ctx->stream));
        if (total == 0) {
            KT_HIP(hipMemsetAsync(d_total, 0, 8, ctx->stream));
        } else {
            hipLaunchKernelGGL(gran_index_kernel, dim3((uint32_t)((n_reads + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                               d_offsets, n_reads, gfirst, n_gran);
            hipLaunchKernelGGL(gran_break_kernel, dim3((uint32_t)((n_gran + BLOCK / 64 - 1) / (BLOCK / 64))), dim3(BLOCK), 0,
                               ctx->stream, d_bases, total, d_offsets, gfirst, n_reads, n_gran, lastbreak);
            if (int rc = device_excl_scan<true>(ctx, lastbreak, n_gran, carry, partial, nullptr)) return rc;
            MinArgs a{d_bases, d_offsets, gfirst, carry, n_reads, total, n_gran,
                      (uint32_t)wsize, (uint32_t)msize, (uint32_t)(wsize - (uint64_t)msize + 1)};
            const bool narrow = msize <= 16;
            uint32_t *act32 = nullptr;
            uint64_t *act64 = nullptr;
            if (two_level) {
                if (int rc = narrow ? wide_window_act<uint32_t>(ctx, a, &act32) : wide_window_act<uint64_t>(ctx, a, &act64)) return rc;
            }
            Chain chain{(unsigned long long *)tcount, (unsigned long long *)(d_total + 1), d_total, n_tiles, capacity};
            if (capacity == 0) {
                // count only: per-tile counts, then their sum
                if (narrow) {
                    auto kern = two_level ? min_tile_kernel<uint32_t, false, 1, 2>
                                          : (hg == 1 ? min_tile_kernel<uint32_t, false, 1> : min_tile_kernel<uint32_t, false, MAX_HG>);
                    hipLaunchKernelGGL(kern, dim3((uint32_t)n_tiles), dim3(BLOCK), 0, ctx->stream, a, tcount, chain,
                                       (Event *)nullptr, (uint8_t *)nullptr, (uint64_t *)nullptr, act32);
                } else {
                    auto kern = two_level ? min_tile_kernel<uint64_t, false, 1, 2>
                                          : (hg == 1 ? min_tile_kernel<uint64_t, false, 1> : min_tile_kernel<uint64_t, false, MAX_HG>);
                    hipLaunchKernelGGL(kern, dim3((uint32_t)n_tiles), dim3(BLOCK), 0, ctx->stream, a, tcount, chain,
                                       (Event *)nullptr, (uint8_t *)nullptr, (uint64_t *)nullptr, act64);
                }
                if (int rc = device_excl_scan<false>(ctx, tcount, n_tiles, tbase, partial, d_total)) return rc;
                KT_HIP(hipMemcpyAsync(&n_ev, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
                KT_HIP(hipStreamSynchronize(ctx->stream));
            } else {
                // single pass: the tiles chain their output offsets themselves (decoupled look-back)
                if (int rc = ctx->s_aux2.reserve(align256(capacity * sizeof(Event)) + capacity + 256)) return rc;
                Event *ev = (Event *)ctx->s_aux2.p;
                uint8_t *ev_type = (uint8_t *)ctx->s_aux2.p + align256(capacity * sizeof(Event));
                KT_HIP(hipMemsetAsync(tcount, 0, n_tiles * 8, ctx->stream));
                KT_HIP(hipMemsetAsync(d_total, 0, 16, ctx->stream));
                if (narrow) {
                    auto kern = two_level ? min_tile_kernel<uint32_t, true, 1, 2>
                                          : (hg == 1 ? min_tile_kernel<uint32_t, true, 1> : min_tile_kernel<uint32_t, true, MAX_HG>);
                    hipLaunchKernelGGL(kern, dim3((uint32_t)n_tiles), dim3(BLOCK), 0, ctx->stream, a, (uint64_t *)nullptr, chain,
                                       ev, ev_type, d_evoff, act32);
                } else {
                    auto kern = two_level ? min_tile_kernel<uint64_t, true, 1, 2>
                                          : (hg == 1 ? min_tile_kernel<uint64_t, true, 1> : min_tile_kernel<uint64_t, true, MAX_HG>);
                    hipLaunchKernelGGL(kern, dim3((uint32_t)n_tiles), dim3(BLOCK), 0, ctx->stream, a, (uint64_t *)nullptr, chain,
                                       ev, ev_type, d_evoff, act64);
                }
                hipLaunchKernelGGL(min_tail_kernel, dim3((uint32_t)((n_reads + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                                   d_evoff, n_reads, d_total);
                KT_HIP(hipGetLastError());
                KT_HIP(hipMemcpyAsync(&n_ev, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
                KT_HIP(hipStreamSynchronize(ctx->stream));
                const uint64_t n_fin = n_ev < capacity ? n_ev : capacity;
                if (n_fin) {
                    hipLaunchKernelGGL(min_finalize_kernel, dim3((uint32_t)((n_fin + 255) / 256)), dim3(256), 0, ctx->stream, ev,
                                       ev_type, n_fin, (uint32_t)wsize, capacity, d_k, d_s, d_e);
                    KT_HIP(hipGetLastError());
                }
            }
        }
        if (total == 0 || capacity == 0) {
            // no emit pass ran: every read's offset is the (possibly zero) total
            hipLaunchKernelGGL(min_tail_kernel, dim3((uint32_t)((n_reads + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                               d_evoff, n_reads, d_total);
            KT_HIP(hipGetLastError());
        }
    }
    *n_events = n_ev;
    const uint64_t n_out = n_ev < capacity ? n_ev : capacity;
    if (mem == KT_MEM_HOST) {
        KT_HIP(hipMemcpyAsync(ev_offsets, d_evoff, (n_reads + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (n_out) {
            KT_HIP(hipMemcpyAsync(kmers, d_k, n_out * 8, hipMemcpyDeviceToHost, ctx->stream));
            KT_HIP(hipMemcpyAsync(starts, d_s, n_out * 8, hipMemcpyDeviceToHost, ctx->stream));
            KT_HIP(hipMemcpyAsync(ends, d_e, n_out * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        KT_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (n_ev > capacity && capacity)
        return kt::fail(KT_ERR_ARG, "kt_minimisers: capacity smaller than the number of minimisers (see *n_events)");
    return KT_OK;
}
