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
// General w: the batch is cut into tiles of 3072 positions (+ a 1024-position halo in front, so
// W <= 1024).  Run starts are a max-scan ("latest break before p") carried across tiles by a
// small prefix pass over 1024-position granules; the sliding minimum is log2(W) doubling steps
// over an LDS array of the tile's m-mers.  Two passes (count, then emit at scanned offsets) keep
// the output dense and in read order; a last pass resolves ws and read-local coordinates.
// HBM traffic is a few bytes per base; the work is LDS/VALU bound.  No MFMA.
#include "kt_internal.hpp"
#include "kt_launch.hpp"

namespace {

constexpr int BLOCK = 256;
constexpr uint32_t GRAN = 1024;               // carry granule (positions)
constexpr uint32_t TILE = 3 * GRAN;           // positions owned by a workgroup
constexpr uint32_t RANGE = TILE + GRAN;       // with the halo granule in front
constexpr uint32_t PER = RANGE / BLOCK;       // 16 consecutive positions per thread
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
// base q gives q + 1), 0 if the granule has none
__global__ __launch_bounds__(BLOCK) void gran_break_kernel(const uint8_t *__restrict__ bases, uint64_t total,
                                                           const uint64_t *__restrict__ offsets,
                                                           const uint64_t *__restrict__ gfirst,
                                                           uint64_t n_reads, uint64_t *__restrict__ lastbreak) {
    __shared__ unsigned long long best;
    const uint64_t g = blockIdx.x;
    if (threadIdx.x == 0) best = 0;
    __syncthreads();
    unsigned long long mine = 0;
    const uint64_t p0 = g * GRAN;
    for (uint32_t i = threadIdx.x; i < GRAN; i += BLOCK) {
        const uint64_t p = p0 + i;
        if (p < total && ktd::nt4(bases[p]) > 3) mine = p + 2;
    }
    for (uint64_t r = gfirst[g] + threadIdx.x; r < n_reads; r += BLOCK) {
        const uint64_t o = offsets[r];
        if (o >= p0 + GRAN) break;
        if (o + 1 > mine) mine = o + 1;
    }
    if (mine) atomicMax(&best, mine);
    __syncthreads();
    if (threadIdx.x == 0) lastbreak[g] = best;
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
    uint64_t val, pos, run;  // minimiser, global position of the event, global start of its run
};

struct TileShared {
    uint64_t a[RANGE];           // m-mers -> doubling -> active minimiser; later the event ranks (u32)
    uint8_t code[RANGE + 32];    // 2-bit codes (4 = ambiguous) of positions range0 - 32 .. range0 + RANGE
    uint32_t start_bits[RANGE / 32 + 1];  // bit per position of the range (+1): a read starts here
    uint64_t scan_tmp[8];
    uint32_t cnt_tmp[8];
};

// exclusive max-scan of one value per thread over the 256-thread workgroup
__device__ __forceinline__ uint64_t block_max_excl(uint64_t v, uint64_t *tmp) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off && o > inc) inc = o;
    }
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    uint64_t base = 0;
    for (uint32_t w = 0; w < wave; w++) base = base > tmp[w] ? base : tmp[w];
    const uint64_t prev = __shfl_up(inc, 1, 64);
    const uint64_t ex = lane ? prev : 0;
    __syncthreads();
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

template <bool EMIT>
__global__ __launch_bounds__(BLOCK) void min_tile_kernel(MinArgs a, uint64_t *__restrict__ tile_count,
                                                         const uint64_t *__restrict__ ev_base, Event *__restrict__ ev,
                                                         uint8_t *__restrict__ ev_type, uint64_t *__restrict__ ev_offsets) {
    __shared__ TileShared sm;
    const uint32_t tid = threadIdx.x;
    const uint64_t t0 = (uint64_t)blockIdx.x * TILE;           // first position owned
    const int64_t range0 = (int64_t)t0 - (int64_t)GRAN;        // position of local index 0 (negative for tile 0)
    const uint32_t m = a.m, w = a.w, W = a.W;

    // ---- stage codes (32 bases of context in front of the range) and read-start bits ----
    for (uint32_t i = tid; i < RANGE + 32; i += BLOCK) {
        const int64_t p = range0 - 32 + (int64_t)i;
        sm.code[i] = (p >= 0 && (uint64_t)p < a.total) ? (uint8_t)ktd::nt4(a.bases[p]) : (uint8_t)4;
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
    auto is_start = [&](uint32_t li) { return (sm.start_bits[li >> 5] >> (li & 31u)) & 1u; };

    // ---- run starts: thread-local pass, workgroup max-scan, carry from the granules before the range ----
    const uint32_t l0 = tid * PER;  // first local index of this thread
    uint64_t local_break = 0;       // 1 + latest run-start candidate among this thread's positions
    for (uint32_t j = 0; j < PER; j++) {
        const uint32_t li = l0 + j;
        const int64_t p = range0 + (int64_t)li;
        if (p < 0 || (uint64_t)p >= a.total) continue;
        if (is_start(li)) local_break = (uint64_t)p + 1;
        if (sm.code[li + 32] > 3) local_break = (uint64_t)p + 2;
    }
    uint64_t brk = block_max_excl(local_break, sm.scan_tmp);
    {
        const uint64_t c = range0 > 0 ? a.carry[(uint64_t)range0 / GRAN] : 0;
        if (c > brk) brk = c;
    }
    // brk - 1 = start of the run that position (range0 + l0 - 1) belongs to, had it been unambiguous

    // ---- canonical m-mers of this thread's positions -> sm.a ----
    {
        uint64_t f = 0, r = 0;
        const uint64_t mask = m == 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
        const uint32_t rsh = 2 * (m - 1);
        // the m - 1 codes in front of the first position (ambiguous ones are harmless: run_len gates the use)
        for (uint32_t j = 0; j + 1 < m; j++) {
            const uint32_t c = sm.code[l0 + 32 - (m - 1) + j] & 3u;
            f = ((f << 2) | c) & mask;
            r = (r >> 2) | ((uint64_t)(3u - c) << rsh);
        }
        uint64_t b = brk;
        for (uint32_t j = 0; j < PER; j++) {
            const uint32_t li = l0 + j;
            const int64_t p = range0 + (int64_t)li;
            const uint32_t cd = sm.code[li + 32];
            const uint32_t c = cd & 3u;
            f = ((f << 2) | c) & mask;
            r = (r >> 2) | ((uint64_t)(3u - c) << rsh);
            uint64_t v = NONE;
            if (p >= 0 && (uint64_t)p < a.total) {
                if (is_start(li)) b = (uint64_t)p + 1;
                if (cd > 3) b = (uint64_t)p + 2;
                const uint64_t run_len = (uint64_t)p + 2 > b ? (uint64_t)p + 2 - b : 0;  // p + 1 - (b - 1)
                if (run_len >= m) v = f < r ? f : r;
            }
            sm.a[li] = v;
        }
    }
    __syncthreads();

    // ---- sliding minimum over W m-mers: doubling, then two overlapping power-of-two windows ----
    {
        uint32_t span = 1;  // sm.a[i] = min of the `span` m-mers ending at i
        while (span * 2 <= W) {
            uint64_t x[PER];
#pragma unroll
            for (uint32_t j = 0; j < PER; j++) {
                const uint32_t li = l0 + j;
                const uint64_t u = sm.a[li];
                const uint64_t o = li >= span ? sm.a[li - span] : NONE;
                x[j] = u < o ? u : o;
            }
            __syncthreads();
#pragma unroll
            for (uint32_t j = 0; j < PER; j++) sm.a[l0 + j] = x[j];
            __syncthreads();
            span *= 2;
        }
        if (span < W) {
            const uint32_t d = W - span;
            uint64_t x[PER];
#pragma unroll
            for (uint32_t j = 0; j < PER; j++) {
                const uint32_t li = l0 + j;
                const uint64_t u = sm.a[li];
                const uint64_t o = li >= d ? sm.a[li - d] : NONE;
                x[j] = u < o ? u : o;
            }
            __syncthreads();
#pragma unroll
            for (uint32_t j = 0; j < PER; j++) sm.a[l0 + j] = x[j];
            __syncthreads();
        }
    }

    // ---- events of the positions this workgroup owns (local index >= GRAN) ----
    uint32_t n_ev = 0;
    uint32_t ev_kind[PER];
    uint64_t ev_val[PER], ev_run[PER];
    {
        uint64_t b = brk;
        for (uint32_t j = 0; j < PER; j++) {
            const uint32_t li = l0 + j;
            const int64_t p = range0 + (int64_t)li;
            ev_kind[j] = 0;
            if (p < 0 || (uint64_t)p >= a.total) continue;
            const uint64_t b_prev = b;  // run bookkeeping as of position p - 1
            const uint32_t cd = sm.code[li + 32];
            const bool st = is_start(li);
            if (st) b = (uint64_t)p + 1;
            if (cd > 3) b = (uint64_t)p + 2;
            if (li < GRAN) continue;  // halo: state only
            const uint64_t run_len = (uint64_t)p + 2 > b ? (uint64_t)p + 2 - b : 0;
            // run length of p - 1 (0 at a read start: the previous base belongs to another read)
            const uint64_t prev_len = (st || p == 0) ? 0 : ((uint64_t)p + 1 > b_prev ? (uint64_t)p + 1 - b_prev : 0);
            const bool last = ((uint64_t)p + 1 == a.total) || is_start(li + 1);
            uint32_t kind = 0;
            uint64_t val = 0, run = 0;
            if (cd > 3) {
                if (prev_len >= w) {  // E2: an ambiguous base closes a full window
                    kind = 2;
                    val = sm.a[li - 1];
                    run = b_prev - 1;
                }
            } else {
                if (run_len > w && sm.a[li] != sm.a[li - 1]) {  // E1: the active minimiser changed
                    kind = 1;
                    val = sm.a[li - 1];
                    run = b - 1;
                } else if (last && run_len >= m) {  // E3: the read's last window
                    kind = 3;
                    val = run_len >= w ? sm.a[li] : NONE;
                    run = b - 1;
                }
            }
            if (kind) {
                ev_kind[j] = kind;
                ev_val[j] = val;
                ev_run[j] = run;
                n_ev++;
            }
        }
    }
    uint32_t tile_total;
    const uint32_t rank0 = block_sum_excl(n_ev, sm.cnt_tmp, &tile_total);
    if (!EMIT) {
        if (tid == 0) tile_count[blockIdx.x] = tile_total;
        return;
    }
    const uint64_t base = ev_base[blockIdx.x];
    // event ranks per owned position (exclusive), for the reads that start inside the tile
    uint32_t *rank = reinterpret_cast<uint32_t *>(sm.a);
    __syncthreads();  // sm.a is done as the minimiser array
    {
        uint32_t rk = rank0;
        for (uint32_t j = 0; j < PER; j++) {
            const uint32_t li = l0 + j;
            rank[li] = rk;
            if (ev_kind[j]) {
                ev[base + rk] = Event{ev_val[j], (uint64_t)(range0 + (int64_t)li), ev_run[j]};
                ev_type[base + rk] = (uint8_t)ev_kind[j];
                rk++;
            }
        }
    }
    __syncthreads();
    for (uint64_t r = a.gfirst[t0 / GRAN] + tid; r < a.n_reads; r += BLOCK) {
        const uint64_t o = a.offsets[r];
        if (o >= t0 + TILE || o >= a.total) break;
        ev_offsets[r] = base + rank[(uint32_t)((int64_t)o - range0)];
    }
}

// reads that start at the very end of the batch (empty, after the last base) were not seen by any tile
__global__ void min_tail_kernel(uint64_t *__restrict__ ev_offsets, uint64_t n_reads, const uint64_t *__restrict__ total) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_reads) return;
    if (r == n_reads || ev_offsets[r] == NONE) ev_offsets[r] = *total;
}

// one thread per read: window starts (previous change of the same run) and read-local coordinates
__global__ void min_finalize_kernel(const uint64_t *__restrict__ offsets, uint64_t n_reads,
                                    const uint64_t *__restrict__ ev_offsets, const Event *__restrict__ ev,
                                    const uint8_t *__restrict__ ev_type, uint32_t w, uint64_t capacity,
                                    uint64_t *__restrict__ kmers, uint64_t *__restrict__ starts,
                                    uint64_t *__restrict__ ends) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t e0 = ev_offsets[r], e1 = ev_offsets[r + 1], rs = offsets[r];
    for (uint64_t j = e0; j < e1 && j < capacity; j++) {
        const Event e = ev[j];
        uint64_t ws = e.run;
        if (j > e0 && ev_type[j - 1] == 1 && ev[j - 1].run == e.run) ws = ev[j - 1].pos + 1 - w;
        kmers[j] = e.val;
        starts[j] = ws - rs;
        ends[j] = (ev_type[j] == 3 ? e.pos + 1 : e.pos) - rs;
    }
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

__global__ void set_last_kernel(uint64_t *__restrict__ ev_offsets, uint64_t n_reads, const uint64_t *__restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ev_offsets[n_reads] = *total;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

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
    if (wsize != 0 && wsize - (uint64_t)msize + 1 > GRAN)
        return kt::fail(KT_ERR_ARG, "kt_minimisers: windows of more than 1024 m-mers are not supported by this build");
    if (n_reads == 0) return KT_OK;
    if (!offsets || !ev_offsets) return kt::fail(KT_ERR_ARG, "kt_minimisers: null offsets");
    if (capacity && (!kmers || !starts || !ends)) return kt::fail(KT_ERR_ARG, "kt_minimisers: null output");
    if (int rc = ctx->use()) return rc;
    uint64_t total = 0;
    if (int rc = total_bases_of(ctx, offsets, n_reads, mem, &total)) return rc;
    if (total && !bases) return kt::fail(KT_ERR_ARG, "kt_minimisers: null bases");

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
    const uint64_t n_tiles = (total + TILE - 1) / TILE;
    const uint64_t n_scan = (n_reads > n_gran ? n_reads : n_gran) + 1;
    size_t off = 0;
    const size_t o_gfirst = off;  off += align256((n_gran + 2) * 8);
    const size_t o_break = off;   off += align256((n_gran + 1) * 8);
    const size_t o_carry = off;   off += align256((n_gran + 1) * 8);
    const size_t o_tcount = off;  off += align256((n_tiles + 1) * 8);
    const size_t o_tbase = off;   off += align256((n_tiles + 1) * 8);
    const size_t o_rcount = off;  off += align256((wsize == 0 ? n_reads + 1 : 1) * 8);
    const size_t o_part = off;    off += align256((n_scan / 1024 + 2) * 8);
    const size_t o_total = off;   off += 256;
    if (int rc = ctx->s_aux1.reserve(off)) return rc;
    char *ib = (char *)ctx->s_aux1.p;
    uint64_t *gfirst = (uint64_t *)(ib + o_gfirst), *lastbreak = (uint64_t *)(ib + o_break);
    uint64_t *carry = (uint64_t *)(ib + o_carry), *tcount = (uint64_t *)(ib + o_tcount);
    uint64_t *tbase = (uint64_t *)(ib + o_tbase), *rcount = (uint64_t *)(ib + o_rcount);
    uint64_t *partial = (uint64_t *)(ib + o_part), *d_total = (uint64_t *)(ib + o_total);

    uint64_t n_ev = 0;
    if (wsize == 0) {
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
            hipLaunchKernelGGL(gran_break_kernel, dim3((uint32_t)n_gran), dim3(BLOCK), 0, ctx->stream, d_bases, total,
                               d_offsets, gfirst, n_reads, lastbreak);
            if (int rc = device_excl_scan<true>(ctx, lastbreak, n_gran, carry, partial, nullptr)) return rc;
            MinArgs a{d_bases, d_offsets, gfirst, carry, n_reads, total, n_gran,
                      (uint32_t)wsize, (uint32_t)msize, (uint32_t)(wsize - (uint64_t)msize + 1)};
            hipLaunchKernelGGL(min_tile_kernel<false>, dim3((uint32_t)n_tiles), dim3(BLOCK), 0, ctx->stream, a, tcount,
                               (const uint64_t *)nullptr, (Event *)nullptr, (uint8_t *)nullptr, (uint64_t *)nullptr);
            if (int rc = device_excl_scan<false>(ctx, tcount, n_tiles, tbase, partial, d_total)) return rc;
            KT_HIP(hipMemcpyAsync(&n_ev, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
            KT_HIP(hipStreamSynchronize(ctx->stream));
            if (capacity && n_ev) {
                if (int rc = ctx->s_aux2.reserve(align256(n_ev * sizeof(Event)) + n_ev + 256)) return rc;
                Event *ev = (Event *)ctx->s_aux2.p;
                uint8_t *ev_type = (uint8_t *)ctx->s_aux2.p + align256(n_ev * sizeof(Event));
                hipLaunchKernelGGL(min_tile_kernel<true>, dim3((uint32_t)n_tiles), dim3(BLOCK), 0, ctx->stream, a,
                                   (uint64_t *)nullptr, tbase, ev, ev_type, d_evoff);
                hipLaunchKernelGGL(min_tail_kernel, dim3((uint32_t)((n_reads + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                                   d_evoff, n_reads, d_total);
                hipLaunchKernelGGL(min_finalize_kernel, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, ctx->stream,
                                   d_offsets, n_reads, d_evoff, ev, ev_type, (uint32_t)wsize, capacity, d_k, d_s, d_e);
                KT_HIP(hipGetLastError());
            }
        }
        if (total == 0 || !(capacity && n_ev)) {
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
