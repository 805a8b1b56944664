// kt_segment.hpp - flat "segment" front-end shared by the k-mer counting, routing and
// debug kernels: canonical k-mers (1 <= k <= 31) of a CSR read batch, independent of
// how long or how many the reads are.
//
// The concatenated bases are cut into fixed segments of SEG = 8192 bases; one
// 256-thread workgroup owns one segment (grid-stride).  Per segment:
//   stage     thread i encodes bases [32i, 32i+32) of the segment (+ one 32-base halo
//             item) with two 16-byte loads into LDS: a 64-bit word of 2-bit codes
//             (first base in the top bits) and a 32-bit mask of invalid bases.
//   bounds    read starts that fall inside the segment are scattered into a third LDS
//             bit mask (found through seg_first[], the per-segment lower bound of the
//             offsets array built by seg_index_kernel) - a k-mer may not span one.
//   k-mers    thread t walks window starts [32t, 32t+32) with a 128-bit shift register
//             (struct Window) - every start position is independent of the reads before
//             it, so no serial scan.
//   sink      a functor receives (fwd, rev, index of the k-mer's last base).
#pragma once
#include "kt_device.hpp"

namespace ktseg {
using ktd::uniform64;

constexpr int BLOCK = 256;
constexpr uint32_t PER_THREAD = 32;
constexpr uint32_t SEG = BLOCK * PER_THREAD;  // 8192 bases
constexpr uint32_t NITEM = BLOCK + 1;         // 32-base items incl. halo

struct SegArgs {
    const uint8_t *bases;
    const uint64_t *offsets;  // n_reads + 1, offsets[0] == 0
    const uint64_t *seg_first;  // n_seg + 1: first read r with offsets[r] >= g * SEG
    uint64_t n_reads;
    uint64_t n_seg;
    uint32_t k;
};

// seg_first[g] = lower_bound(offsets[0..n_reads], g * SEG) for g in [0, n_seg]
static __global__ void seg_index_kernel(const uint64_t *__restrict__ offsets, uint64_t n_reads,
                                 uint64_t *__restrict__ seg_first, uint64_t n_seg) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_reads) return;
    const uint64_t cur = offsets[r];
    uint64_t g_lo = (r == 0) ? 0 : offsets[r - 1] / SEG + 1;
    uint64_t g_hi = cur / SEG;
    if (r == n_reads) g_hi = n_seg;  // everything past the end maps to the sentinel
    if (g_hi > n_seg) g_hi = n_seg;
    for (uint64_t g = g_lo; g <= g_hi; g++) seg_first[g] = r;
}

struct SegShared {
    uint64_t codes[NITEM + 1];
    uint32_t inv[NITEM + 1];
    uint32_t bnd[NITEM + 1];
};

// Stages segment `g` into LDS (codes, invalid-base mask, read-boundary mask).  Must be called by
// all 256 threads; ends with a barrier.  The caller needs another barrier before the next
// stage_segment() reuses the LDS (for_each_kmer / collect_kmers do that themselves).
// (`tid` = the thread's index among the 256 that stage this segment: a workgroup of several 256-thread groups
// stages one segment per group, every group into its own SegShared; the barriers are the whole workgroup's.)
__device__ __forceinline__ void stage_segment(const SegArgs &a, uint64_t g, SegShared &sm, const uint32_t tid) {
    const uint64_t total = a.offsets[a.n_reads];
    const uint64_t B0 = g * SEG;

    // ---- stage: 32 bases per item ------------------------------------------------
    for (uint32_t i = tid; i < NITEM; i += BLOCK) {
        const uint64_t b = B0 + 32ull * i;
        uint64_t w = 0;
        uint32_t iv = 0xFFFFFFFFu;
        if (b < total) {
            uint32_t d[8];
            if (b + 32 <= total) {
                __builtin_memcpy(d, a.bases + b, 32);
            } else {
                unsigned char raw[32];
                for (int j = 0; j < 32; j++) raw[j] = (b + j < total) ? a.bases[b + j] : (unsigned char)'N';
                __builtin_memcpy(d, raw, 32);
            }
            // four bases per swar4 (letters only); an item holding raw bytes 0..3 takes the per-byte path
            uint32_t msb_first = 0, any_raw = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                uint32_t c8, i4, rw;
                ktd::swar4(d[q], c8, i4, rw);
                w = (w << 8) | c8;
                msb_first = (msb_first << 4) | i4;
                any_raw |= rw;
            }
            iv = __builtin_bitreverse32(msb_first);  // bit j = base j is not a nucleotide
            if (any_raw) {
                w = 0;
                iv = 0;
                for (int j = 0; j < 32; j++) {
                    const uint32_t e = ktd::nt4((d[j >> 2] >> (8 * (j & 3))) & 0xFFu);
                    w = (w << 2) | (e & 3u);
                    iv |= (e >> 2) << j;
                }
            }
        }
        sm.codes[i] = w;
        sm.inv[i] = iv;
        sm.bnd[i] = 0;
    }
    ktd::lds_barrier();

    // ---- read boundaries inside (B0, B0 + SEG + 32) ---------------------------------
    {
        const uint64_t lim = B0 + SEG + 32;
        for (uint64_t r = a.seg_first[g] + tid; r < a.n_reads; r += BLOCK) {
            const uint64_t o = a.offsets[r];
            if (o >= lim) break;
            const uint32_t rel = (uint32_t)(o - B0);
            atomicOr(&sm.bnd[rel >> 5], 1u << (rel & 31u));
        }
    }
    ktd::lds_barrier();
}

__device__ __forceinline__ void stage_segment(const SegArgs &a, uint64_t g, SegShared &sm) {
    stage_segment(a, g, sm, threadIdx.x);
}

// ---- the same staging with EVERY global read of a unit requested a unit ahead (kt_bulk.hip's level 1, scatter1x) -------------
// stage_segment waits for two dependent global reads in a row - the bases, then (behind a barrier) seg_first -> offsets -
// while the workgroup has nothing else to do: ~3.5 us per 8192-base unit.  Round 4 requested the bases a unit ahead, but
// still started a unit with a read the thread has to wait for - its first read start, offsets[seg_first[g] + tid] -
// and that read is YOUNGER than the stores of the previous unit's copy-out: a wave's memory operations complete in issue
// order, so the wait drains the wave's whole store queue at every unit (the k = 7 lesson of kt_oligo.hip); and the
// compiler waits vmcnt(0) for the prefetched bases as well - it cannot count the stores issued behind them through the
// copy-out's control flow.  Here a unit's prefetch carries the bases of unit g, the thread's first read start of unit g
// (seg_first[g] is known: it came with the prefetch before) and seg_first of the unit after - requested BEFORE the
// copy-out's stores, from inline assembly (the compiler inserts no wait for what it does not know about), and taken
// behind them with s_waitcnt vmcnt(number of stores): every read has landed, the stores stay in flight.
// Rules (the same as kt_bulk.hip's buf_load8_async / buf_take; tools/check_inflight.py reads the assembly at build
// time): nothing touches the in-flight registers between prefetch_issue and prefetch_take - a spill or a reload of one of
// them there would move garbage (scratch elsewhere in the kernel is harmless: csrc/Makefile) -; at least `NSTORE` vector
// memory operations are issued between the two on every path (the copy-out stores unconditionally: a group with nothing to
// write goes to a dump line - the checker counts the instructions of the layout, the GPU suite checks the paths).
struct SegHalo { uint32_t d[8]; };
struct SegPrefetch2 {
    uint64_t a[4];         // item tid of unit g: bases [32 tid, 32 tid + 32) - in flight until prefetch_take
    uint64_t o0;           // offsets[first(g) + tid] - in flight
    SegHalo halo;          // the halo item (BLOCK): the same for the whole group, read through the scalar cache
    uint64_t first_next;   // seg_first of the unit that will be prefetched next (scalar cache)
    bool whole0, whole1, has_o;  // the item lies wholly inside the batch / the read exists (else: not loaded)
};

// (`safe`: 32 readable bytes - a lane that has nothing to read reads them instead, so that the loads are issued
// unconditionally: an asm statement under a condition makes its outputs phi nodes, which the allocator copies at will)
__device__ __forceinline__ void prefetch_issue(SegPrefetch2 &pf, const SegArgs &a, uint64_t g, uint64_t first_g,
                                               uint64_t g_next, const uint32_t tid, const void *safe) {
    const uint64_t total = ktd::load_uniform(a.offsets + a.n_reads);
    const uint64_t gu = uniform64(g);  // (the unit is the same for the 256 threads of a group)
    const uint64_t b0 = gu * SEG + 32ull * tid, b1 = gu * SEG + 32ull * BLOCK;
    pf.whole0 = b0 + 32 <= total;
    pf.whole1 = b1 + 32 <= total && ((uintptr_t)a.bases & 3u) == 0;  // (a scalar read wants a dword address)
    const uint64_t r0 = first_g + tid;
    pf.has_o = r0 < a.n_reads;
    pf.first_next = ktd::load_uniform(a.seg_first + uniform64(g_next));
#pragma unroll
    for (int q = 0; q < 8; q++) pf.halo.d[q] = 0;
    if (pf.whole1) {
        typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
        const u32x8 v = ktd::load_uniform(reinterpret_cast<const u32x8 *>(a.bases + b1));
#pragma unroll
        for (int q = 0; q < 8; q++) pf.halo.d[q] = v[q];
    }
    const void *p0 = pf.whole0 ? (const void *)(a.bases + b0) : safe;
    const void *po = pf.has_o ? (const void *)(a.offsets + r0) : safe;
    asm volatile("global_load_dwordx2 %0, %5, off\n\tglobal_load_dwordx2 %1, %5, off offset:8\n\t"
                 "global_load_dwordx2 %2, %5, off offset:16\n\tglobal_load_dwordx2 %3, %5, off offset:24\n\t"
                 "global_load_dwordx2 %4, %6, off"
                 : "=&v"(pf.a[0]), "=&v"(pf.a[1]), "=&v"(pf.a[2]), "=&v"(pf.a[3]), "=&v"(pf.o0) : "v"(p0), "v"(po) : "memory");
}

// what prefetch_issue requested, once at most NSTORE vector memory operations issued behind it are outstanding
struct SegTaken {
    uint32_t d0[8];
    SegHalo halo;
    uint64_t o0, first_next;
    bool whole0, whole1;
};
template <int NSTORE>
__device__ __forceinline__ void prefetch_take(SegTaken &tk, SegPrefetch2 &pf) {
    uint64_t a[4], o0;
    // (the in-flight registers are inputs only: with in-out operands the allocator is free to copy them in front of the wait)
    asm volatile("s_waitcnt vmcnt(%10)\n\tv_mov_b64 %0, %5\n\tv_mov_b64 %1, %6\n\tv_mov_b64 %2, %7\n\tv_mov_b64 %3, %8\n\tv_mov_b64 %4, %9"
                 : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(o0)
                 : "v"(pf.a[0]), "v"(pf.a[1]), "v"(pf.a[2]), "v"(pf.a[3]), "v"(pf.o0), "n"(NSTORE)
                 : "memory");
#pragma unroll
    for (int q = 0; q < 4; q++) {
        tk.d0[2 * q] = (uint32_t)a[q];
        tk.d0[2 * q + 1] = (uint32_t)(a[q] >> 32);
    }
    tk.halo = pf.halo;
    tk.o0 = pf.has_o ? o0 : ~0ull;
    tk.first_next = pf.first_next;
    tk.whole0 = pf.whole0;
    tk.whole1 = pf.whole1;
}

// the 32 bases at batch position b (din: their bytes when the item lies wholly inside the batch) as a word of 2-bit codes,
// first base on top, and the mask of the bases that are not nucleotides (all ones past the batch's end)
__device__ __forceinline__ void encode_item(const SegArgs &a, uint64_t total, uint64_t b, const uint32_t (&din)[8], bool whole,
                                            uint64_t &w, uint32_t &iv) {
    w = 0;
    iv = 0xFFFFFFFFu;
    if (b < total) {
        uint32_t d[8];
#pragma unroll
        for (int q = 0; q < 8; q++) d[q] = din[q];
        if (!whole) {  // the batch ends inside this item
            unsigned char raw[32];
            for (int j = 0; j < 32; j++) raw[j] = (b + j < total) ? a.bases[b + j] : (unsigned char)'N';
            __builtin_memcpy(d, raw, 32);
        }
        uint32_t msb_first = 0, any_raw = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            uint32_t c8, i4, rw;
            ktd::swar4(d[q], c8, i4, rw);
            w = (w << 8) | c8;
            msb_first = (msb_first << 4) | i4;
            any_raw |= rw;
        }
        iv = __builtin_bitreverse32(msb_first);
        if (any_raw) {
            w = 0;
            iv = 0;
            for (int j = 0; j < 32; j++) {
                const uint32_t e = ktd::nt4((d[j >> 2] >> (8 * (j & 3))) & 0xFFu);
                w = (w << 2) | (e & 3u);
                iv |= (e >> 2) << j;
            }
        }
    }
}

__device__ __forceinline__ void stage_taken(const SegArgs &a, uint64_t g, uint64_t first_g, SegShared &sm,
                                            const uint32_t tid, const SegTaken &pf) {
    const uint64_t total = ktd::load_uniform(a.offsets + a.n_reads);
    const uint64_t B0 = g * SEG;
    auto encode = [&](uint32_t i, const uint32_t (&din)[8], bool whole) {
        uint64_t w;
        uint32_t iv;
        encode_item(a, total, B0 + 32ull * i, din, whole, w, iv);
        sm.codes[i] = w;
        sm.inv[i] = iv;
        sm.bnd[i] = 0;
    };
    encode(tid, pf.d0, pf.whole0);
    if (tid + BLOCK < NITEM) encode(tid + BLOCK, pf.halo.d, pf.whole1);
    ktd::lds_barrier();
    {
        const uint64_t lim = B0 + SEG + 32, r0 = first_g + tid;
        if (r0 < a.n_reads && pf.o0 < lim) {
            const uint32_t rel = (uint32_t)(pf.o0 - B0);
            atomicOr(&sm.bnd[rel >> 5], 1u << (rel & 31u));
            // more than 256 read starts in a segment (reads below 32 bases): read here, where no loaded value leaves the
            // loop - one that did would put a wait for every outstanding memory operation behind the loop, on every path
            for (uint64_t r = r0 + BLOCK; r < a.n_reads; r += BLOCK) {
                const uint64_t o = a.offsets[r];
                if (o >= lim) break;
                const uint32_t rel2 = (uint32_t)(o - B0);
                atomicOr(&sm.bnd[rel2 >> 5], 1u << (rel2 & 31u));
            }
        }
    }
    ktd::lds_barrier();
}

// The same a segment ahead with PLAIN loads (route_kernel, pack_segments_kernel: kernels whose stores are long gone by the
// time the next segment is staged, so the compiler's own wait costs nothing): request_ahead right behind the staging of
// segment g asks for segment g_next's reads, stage_ahead stages them when their turn comes.
struct SegAhead {
    uint32_t d0[8];
    SegHalo halo;
    uint64_t o0, first, first_next;
    bool whole0, whole1;
};
// g: the segment asked for (the same for the 256 threads), first_g = seg_first[g], g_after: the segment that will be asked
// for next (its seg_first comes along)
__device__ __forceinline__ SegAhead request_ahead(const SegArgs &a, uint64_t g, uint64_t first_g, uint64_t g_after, const uint32_t tid) {
    SegAhead p;
    const uint64_t total = ktd::load_uniform(a.offsets + a.n_reads);
    const uint64_t gu = uniform64(g);
    const uint64_t b0 = gu * SEG + 32ull * tid, b1 = gu * SEG + 32ull * BLOCK;
    p.whole0 = b0 + 32 <= total;
    p.whole1 = b1 + 32 <= total && ((uintptr_t)a.bases & 3u) == 0;  // (a scalar read wants a dword address)
    p.first = first_g;
    p.first_next = ktd::load_uniform(a.seg_first + uniform64(g_after < a.n_seg ? g_after : a.n_seg));
#pragma unroll
    for (int q = 0; q < 8; q++) p.d0[q] = p.halo.d[q] = 0;
    if (p.whole1) {
        typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
        const u32x8 v = ktd::load_uniform(reinterpret_cast<const u32x8 *>(a.bases + b1));
#pragma unroll
        for (int q = 0; q < 8; q++) p.halo.d[q] = v[q];
    }
    if (p.whole0) __builtin_memcpy(p.d0, a.bases + b0, 32);
    const uint64_t r0 = first_g + tid;
    p.o0 = r0 < a.n_reads ? a.offsets[r0] : ~0ull;
    return p;
}
// stages segment g from what request_ahead(.., g, ..) brought (ends with a barrier, like stage_segment)
__device__ __forceinline__ void stage_ahead(const SegArgs &a, uint64_t g, SegShared &sm, const uint32_t tid, const SegAhead &ah) {
    SegTaken tk;
#pragma unroll
    for (int q = 0; q < 8; q++) tk.d0[q] = ah.d0[q];
    tk.halo = ah.halo;
    tk.o0 = ah.o0;
    tk.first_next = ah.first_next;
    tk.whole0 = ah.whole0;
    tk.whole1 = ah.whole1;
    stage_taken(a, g, ah.first, sm, tid, tk);
}

// The thread's walk over its 32 window starts [32*tid, 32*tid + 32) of a staged segment: a
// 128-bit shift register held in two VGPR pairs: fwd = top 2k bits, rev rolls like the
// reference's generator (kmer/src/kmer.rs:91-93).
struct Window {
    uint64_t hi, lo, f, r;
    uint32_t sh, rsh, okm;
    // OR of x >> i for i in [0, len): log2(len) shift-or steps (len is the same for every lane)
    static __device__ __forceinline__ uint64_t or_span(uint64_t x, uint32_t len) {
        uint32_t c = 1;
        while (2 * c <= len) {
            x |= x >> c;
            c *= 2;
        }
        if (c < len) x |= x >> (len - c);
        return x;
    }
    __device__ __forceinline__ Window(const SegShared &sm, uint32_t tid, uint32_t k) {
        hi = sm.codes[tid];
        lo = sm.codes[tid + 1];
        const uint64_t iv = (uint64_t)sm.inv[tid] | ((uint64_t)sm.inv[tid + 1] << 32);
        const uint64_t bd = (uint64_t)sm.bnd[tid] | ((uint64_t)sm.bnd[tid + 1] << 32);
        sh = 64u - 2u * k;
        rsh = 2u * (k - 1);
        // which of the 32 window starts are k-mers, all at once: start j is one iff none of bases j .. j + k - 1 is
        // invalid and no read starts at j + 1 .. j + k - 1 (testing every start on its own was 8 instructions x 32)
        const uint64_t bad = or_span(iv, k) | (k > 1 ? or_span(bd >> 1, k - 1) : 0ull);
        okm = ~(uint32_t)bad;
        f = hi >> sh;
        r = ktd::rev_comp(f, (int)k);
    }
    // the same from the words themselves (two packed items of kt_bulk.hip's PackedSource): iv / bd = the invalid-base and
    // read-start masks of the 64 bases
    __device__ __forceinline__ Window(uint64_t hi_, uint64_t lo_, uint64_t iv, uint64_t bd, uint32_t k) {
        hi = hi_;
        lo = lo_;
        sh = 64u - 2u * k;
        rsh = 2u * (k - 1);
        const uint64_t bad = or_span(iv, k) | (k > 1 ? or_span(bd >> 1, k - 1) : 0ull);
        okm = ~(uint32_t)bad;
        f = hi >> sh;
        r = ktd::rev_comp(f, (int)k);
    }
    // the same walk over bases that are already 2-bit codes (a record of kt_superkmer.hpp): up to 64 bases in (hi_, lo_),
    // the first in the top bits of hi_; bit j of okm_ = window start j is a k-mer
    __device__ __forceinline__ Window(uint64_t hi_, uint64_t lo_, uint32_t okm_, uint32_t k) {
        hi = hi_;
        lo = lo_;
        sh = 64u - 2u * k;
        rsh = 2u * (k - 1);
        okm = okm_;
        f = hi >> sh;
        r = ktd::rev_comp(f, (int)k);
    }
    // is window start j (0..31) a k-mer: k valid bases, no read start among the k-1 later ones
    __device__ __forceinline__ bool ok(uint32_t j) const { return (okm >> j) & 1u; }
    // slide one base: the next code enters fwd at the bottom, its complement enters rev at the top
    __device__ __forceinline__ void step() {
        hi = (hi << 2) | (lo >> 62);
        lo <<= 2;
        f = hi >> sh;
        r = (r >> 2) | ((uint64_t)(3u - (uint32_t)(f & 3u)) << rsh);
    }
};

// Runs `sink(fwd, rev, end_index)` for every valid k-mer of segment `g`.
// Must be called by all 256 threads of the workgroup.
template <class Sink>
__device__ __forceinline__ void for_each_kmer(const SegArgs &a, uint64_t g, SegShared &sm, Sink &&sink) {
    stage_segment(a, g, sm);
    {
        Window w(sm, threadIdx.x, a.k);
        const uint64_t end0 = g * SEG + 32ull * threadIdx.x + (a.k - 1);
#pragma unroll 4
        for (uint32_t j = 0; j < PER_THREAD; j++) {
            if (w.ok(j)) sink(w.f, w.r, end0 + j);
            w.step();
        }
    }
    ktd::lds_barrier();  // LDS is reused by the next segment
}

// Same walk, but the thread's 32 canonical k-mers stay in registers: keys[j] = min(fwd, rev) of
// window start 32*tid + j, bit j of `ok` says whether that window is a valid k-mer.  Fully
// unrolled so the array is register-allocated; lets a caller make several passes over the
// segment's k-mers (count, place) after a single front-end pass.
__device__ __forceinline__ void collect_kmers(const SegArgs &a, uint64_t g, SegShared &sm, uint64_t (&keys)[PER_THREAD],
                                              uint32_t &ok_mask) {
    stage_segment(a, g, sm);
    Window w(sm, threadIdx.x, a.k);
    ok_mask = 0;
#pragma unroll
    for (uint32_t j = 0; j < PER_THREAD; j++) {
        ok_mask |= (w.ok(j) ? 1u : 0u) << j;
        keys[j] = w.f < w.r ? w.f : w.r;
        w.step();
    }
    ktd::lds_barrier();  // LDS is reused by the next segment
}

}  // namespace ktseg
