// kt_bulk.hip - adding a whole batch of k-mers to the table without global atomics: partition by hash prefix, then
// build (or rebuild) every range of the table in LDS.
//
// Why: global atomics on this chip are capped at ~27 G/s wherever they land
// (tools/ubench/atomic_region.hip, profiles/r1_ctr_notes.txt), so the probing path
// (one CAS or atomic add per k-mer, kt_ctr.hip) cannot exceed ~20 G k-mers/s.  A batch can instead go through
// streaming passes (a job: kt_bulk_begin, level 1 over one or more sources, kt_bulk_finish):
//
//   scatter1x persistent workgroups (one per CU, four 256-thread groups) run a source's units (8192-base segments of
//             reads, or 8192 keys of an array that another GPU routed here), four at a time; a round's 16 K keys
//             (32 K 32-bit ones) are counting-sorted in LDS by d1 = top b1 bits of khash(key) and every d1 run is
//             appended to the bucket's fixed region of the key array - its place taken with one atomic per run on the
//             cursor the workgroups of an XCD share (whole cache lines leave the L2), written in 16-byte groups.
//             64-bit keys are stored as khash(key) from here on (to_stored / from_stored).
//   part2     one workgroup per level-1 bucket: the bucket is read in 16384-key chunks, each chunk
//             counting-sorted in LDS by d2 (next b2 hash bits) and its runs appended to the fine buckets, which
//             own fixed shares of the bucket's region; a bucket whose keys do not spread that evenly is noticed
//             during the pass and redone with exact fine boundaries (see part2_kernel)
//   build     one workgroup per fine bucket (d1,d2) = one range of the table (kt_table.hpp): its keys are inserted
//             into the range's image in LDS - on top of what the range already holds when the table has data - and
//             the range is written out: its occupied entries only, packed (a fresh table: the "dense" state), or the
//             whole image with 16-byte coalesced stores, empty slots included (see build_kernel)
//
// Fixed regions assume the hash spreads the batch; a batch dominated by a few k-mers overflows a level-1 region,
// is noticed (one 4-byte read by the host), and is redone from its sources with exact offsets:
//   hist1     a front-end pass that counts k-mers per (workgroup, d1)          (LDS counters)
//   scan1     exact output offset of every (workgroup, d1) pair
//   scatter1  one 256-thread workgroup per unit, every run copied to its exact place
//   part2     first histograms the whole bucket by d2 (a second read) to get the fine boundaries
//
// Traffic ~ 1 B/base + 8 B/k-mer x 4 + 12 B per distinct k-mer (dense) or 16 B/slot (image), all streaming.  The
// intermediate key arrays and the LDS sort / insert arrays hold 32-bit keys when k <= 16 (template parameter K): half
// the partition traffic, and 32-bit LDS atomics in build.
#include <stdio.h>
#include <stdlib.h>

#include <new>
#include <type_traits>
#include <vector>

#include "kt_launch.hpp"
#include "kt_segment.hpp"
#include "kt_superkmer.hpp"
#include "kt_table.hpp"

namespace {

using ktseg::SegArgs;
using ktseg::SegShared;
using kttab::Slot;
using kttab::TableRef;

constexpr int BLOCK = ktseg::BLOCK;       // 256
#ifndef KT_BUILD_T
#define KT_BUILD_T 1024
#endif
#ifndef KT_ABLATION
#define KT_ABLATION 0  // 1 (tools/build_variant*.sh only): KT_BUILD_DBG bits can switch phases of build_kernel off for
#endif                 // profiling.  The shipped library has no switch that skips work.
constexpr uint32_t LOG2_S = kttab::LOG2_RANGE;  // hash positions per fine bucket = per range of the table
constexpr uint32_t S = 1u << LOG2_S;            // 8192 positions: 1024 * m8 slots, 80 - 128 KB of table
constexpr uint32_t MAX_B = 2048;          // digits per level (11 bits)
// part2's shape per key type.  The pass is bound by the short runs it scatters (one per fine bucket and chunk; PMC and
// sweeps in profiles/r2_ctr_k31_pmc.txt), so 64-bit keys take chunks of 16384 keys = 128 KB of LDS in one 1024-thread
// workgroup per CU - the 2-byte digit that used to sit beside every sorted key is hashed again instead, which is what
// makes the room (ctr k=31: 19.5 -> 16.2 ms).  32-bit keys already sort 16384 at a time with 512 threads (the wider
// shape measured 6 % slower there).
// (BIG = that shape; it needs B2 <= 1024 to fit 160 KB, otherwise - and for 32-bit keys - the 512-thread shape runs)
#ifndef KT_STORE_HASH
#define KT_STORE_HASH 1  // 64-bit keys travel through the partition passes as khash(key) (to_stored / from_stored below)
#endif
#ifndef KT_P2_BIG32_PER
#define KT_P2_BIG32_PER 16  // keys per thread of the 1024-thread shape with 32-bit keys (KT_P2_BIG32=1; measured at k=15:
#endif                      // 18.8 ms with 16, 33.2 ms with 32, against 18.3 ms for the 512-thread shape - off by default)
#ifndef KT_P2_T64
#define KT_P2_T64 512  // threads of the BIG shape with 64-bit keys (16384-key chunks either way): 512 x 32 keys per thread
#endif                 // in 256 registers, or 1024 x 16 in 128 (round 3: 1024.  Round 4's batched LDS phases and
                       // in-flight stores want the registers: at 128 the hot loop spilled, and a scratch reload is a vmcnt(0))
template <class K, bool BIG>
constexpr int p2t() { return BIG ? (sizeof(K) == 8 ? KT_P2_T64 : 1024) : 512; }  // threads of a part2 workgroup
template <class K, bool BIG>
constexpr bool p2_sdig() { return !BIG && !(sizeof(K) == 8 && KT_STORE_HASH); }  // digit kept beside every sorted key
                                                           // (not when it is a bit field of the stored hash)
// keys sorted at a time in part2
template <class K, bool BIG>
constexpr uint32_t chunk2() { return sizeof(K) == 8 ? (BIG ? 16384u : 8192u) : (BIG ? (uint32_t)KT_P2_BIG32_PER : 32u) * p2t<K, BIG>(); }

struct Plan {
    uint32_t n;       // hash bits that address the table: cap = m8 * 2^(n-3) (kttab::Geom)
    uint32_t m8;
    uint32_t b1, b2;  // hash bits per level; b1 + b2 + LOG2_S == n
    uint32_t B1, B2;
    uint32_t G;       // persistent workgroups of hist1 / scatter1
    uint64_t cap1;    // paged level 1: keys of room per level-1 bucket in the level-1 output (keys1)
    uint64_t cap2;    // ... and per fine bucket in keys2 (room1 / B2)
    uint64_t room1;   // keys of room per level-1 bucket in keys2 (= cap1)
    uint32_t nxs, xmap;   // level 1 with per-XCD cursors (scatter1x): log2 of the cursor sets, and which set a hardware
                          // XCC_ID appends to (eight nibbles)
    uint32_t dbg;     // KT_BUILD_DBG: ablation switches of build_kernel (profiling only)
    uint32_t lists;   // the dense build's LDS has room for the claim lists behind the image (build_kernel)
    uint32_t kbits;   // 2k for k <= 16 (the table's hash is the bijection ktd::nhash), else 0
};

template <class K>
constexpr K empty_of() { return (K) ~(K)0; }  // K = uint64_t: KT_EMPTY_KEY; canonical k-mers of k <= 16 are never all ones

struct Meta {           // device arrays carved from ctr->b_meta
    uint32_t *H;        // [G][B1] k-mers of workgroup g in bucket d1
    uint64_t *O;        // [G][B1] global offset where workgroup g writes its d1 keys
    uint64_t *bstart;   // [B1 + 1] level-1 bucket boundaries in keys1 (bstart[B1] = #k-mers)
    uint64_t *gcur;     // [B1] paged level 1: the bucket's extent, written by xcd_tails_kernel - what level 2 reads up to
    unsigned long long *dump;  // [1024][2] where scatter1x's copy-out sends the stores that have nothing to write
    unsigned long long *xcur;  // [8][B1] per-XCD cursors: keys XCD set x has appended to bucket d (scatter1x)
    uint32_t *ovf;      // [0]  paged level 1: a bucket ran out of room
    kt_seg_src *srcs;   // [1] part2's source (device copy)
    uint32_t *fail;     // [B1] part2_fast_kernel: the bucket did not fit its fixed fine regions
    uint64_t *fstart;   // [B1 * B2] fine buckets in keys2: [fstart, fend)
    uint64_t *fend;     // [B1 * B2]
    uint64_t *spill_n;  // [1]
    uint64_t *spill_keys;
    uint32_t *spill_counts;
    uint64_t spill_cap;
};

// (both digits lie in the hash's high word - 1 <= b1, b1 + b2 <= 21: plan_job - so they are 32-bit shifts; written on the 64-bit
// value, with a shift the compiler does not know to be >= 32, each is a 64-bit shift: per key, in every pass)
__device__ __forceinline__ uint32_t digit1h(uint64_t h, const Plan &p) { return (uint32_t)(h >> 32) >> (32 - p.b1); }
__device__ __forceinline__ uint32_t digit2h(uint64_t h, const Plan &p) {
    return ((uint32_t)(h >> 32) >> (32 - p.b1 - p.b2)) & (p.B2 - 1);
}
__device__ __forceinline__ uint32_t digit1(uint64_t key, const Plan &p) { return digit1h(ktd::khash_k(key, p.kbits), p); }
// What travels through the partition passes ("stored form" of a canonical k-mer): 64-bit keys travel as
// khash(key) - the hash is a bijection (an odd multiply and an xor of the high half into the low half), so the
// digits and the home slot of every later pass are bit fields of what is already there and the key comes back once,
// where build writes the table (the hash used to be evaluated seven times per k-mer); 32-bit keys (k <= 16) travel
// as they are (their 64-bit hash would not fit).  The empty marker stays all ones in both forms: the only key
// that hashes to it is 0x66c88cc300000000, not a k-mer of k <= 31.
static_assert(KT_KHASH == 1 || !KT_STORE_HASH,
              "the empty marker must not be the hash of a k-mer (true for the multiplicative hash)");
template <class K>
constexpr bool stores_hash() { return sizeof(K) == 8 && KT_STORE_HASH; }
template <class K>
__device__ __forceinline__ K to_stored(uint64_t key) {
    if constexpr (stores_hash<K>()) return (K)ktd::khash(key);
    else return (K)key;
}
template <class K>
__device__ __forceinline__ uint64_t hash_of_stored(K s, const Plan &p) {
    if constexpr (stores_hash<K>()) return (uint64_t)s;
    else if constexpr (sizeof(K) == 4) return (uint64_t)ktd::nhash_top((uint32_t)s, p.kbits) << 32;  // (k <= 16: kbits = 2k, never 0 -
    else return ktd::khash_k((uint64_t)s, p.kbits);  // khash_k's test of it is a branch per key and pass: plan_job)
}
template <class K>
__device__ __forceinline__ uint64_t from_stored(K s) {
    if constexpr (stores_hash<K>()) return ktd::khash_inv((uint64_t)s);
    else return (uint64_t)s;
}

// exclusive prefix sum of cnt[0..B) into out[0..B) (LDS arrays), B <= MAX_B; returns the total.
// All 256 threads must call; tmp is a 256-entry LDS scratch.  Thread sums are scanned inside
// each wave with shuffles and across the 4 waves through tmp: 2 barriers.
template <int NT = BLOCK>
__device__ __forceinline__ uint32_t block_excl_scan(const uint32_t *cnt, uint32_t *out, uint32_t B, uint32_t *tmp) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t per = (B + NT - 1) / NT;  // <= 8
    const uint32_t lo = tid * per;
    uint32_t sum = 0;
    for (uint32_t i = 0; i < per; i++)
        if (lo + i < B) sum += cnt[lo + i];
    uint32_t inc = sum;  // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += v;
    }
    if (lane == 63) tmp[wave] = inc;
    ktd::lds_barrier();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < NT / 64; w++) {
        const uint32_t t = tmp[w];
        if (w < wave) wbase += t;
        total += t;
    }
    uint32_t run = wbase + inc - sum;  // exclusive prefix of this thread's group
    for (uint32_t i = 0; i < per; i++) {
        if (lo + i < B) {
            const uint32_t c = cnt[lo + i];
            out[lo + i] = run;
            run += c;
        }
    }
    ktd::lds_barrier();
    return total;
}

// ---- vector memory that the compiler's s_waitcnt bookkeeping does not see ---------------------------------------
// The partition kernels prefetch the next chunk's keys and then scatter the current chunk's runs; the wave must wait for
// the loads before the next count, and must NOT wait for the stores - they should drain while the next chunk is counted,
// scanned and placed.  vmcnt retires in order, so "the loads are in" is vmcnt(number of stores issued after them).  The
// compiler can only emit that when it can count the stores on every path into the wait; with the first chunk loaded
// ahead of the loop, stores under exec-mask branches and the odd scratch reload it falls back to vmcnt(0) - every wave
// drained its stores once per chunk (ablation: part2 13.8 ms, 6.7 ms without its stores).  So the loads and stores of
// the chunk loops are issued from inline assembly (the compiler inserts no waits for what it does not know about) and
// waited for by hand: buf_wait<N>() = s_waitcnt vmcnt(N), tied to the loaded registers so that no use is moved above it.
// Rules that keep this sound: a loaded value is touched by nothing between its load and its buf_wait (same iteration,
// no loop-carried copy of a value that has not arrived); a store that must not happen gets an offset outside the
// resource (the hardware drops it), so the number of stores per chunk is fixed; no scratch traffic inside the loops
// (checked in the ISA: a spill of an in-flight register would save garbage).
typedef int kt_i32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t BUF_DROP = 0x80000000u;  // an offset beyond any resource made here (all are shorter than 2^31 bytes)
__device__ __forceinline__ kt_i32x4 buf_rsrc(const void *base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    kt_i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xFFFFu));  // (stride 0: raw buffer)
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);                           // num_records = bytes
    r.w = 0x00020000;                                                           // 32-bit data format, no swizzle
    return r;
}
// eight loads: d[j] = the K at byte offset voff + j * STRIDE of the resource; voff moves on by 8 * STRIDE.  (One offset
// register bumped between the loads: per-load scalar offsets were thirty-two more scalar registers than the kernels have.)
template <class K, int STRIDE>
__device__ __forceinline__ void buf_load8_async(K *d, kt_i32x4 rs, uint32_t &voff) {
#define KT_LD8(INSN)                                                                                                      \
    asm volatile(INSN " %0, %8, %9, 0 offen\n\tv_add_u32 %8, %10, %8\n\t" INSN " %1, %8, %9, 0 offen\n\tv_add_u32 %8, %10, %8\n\t" \
                 INSN " %2, %8, %9, 0 offen\n\tv_add_u32 %8, %10, %8\n\t" INSN " %3, %8, %9, 0 offen\n\tv_add_u32 %8, %10, %8\n\t" \
                 INSN " %4, %8, %9, 0 offen\n\tv_add_u32 %8, %10, %8\n\t" INSN " %5, %8, %9, 0 offen\n\tv_add_u32 %8, %10, %8\n\t" \
                 INSN " %6, %8, %9, 0 offen\n\tv_add_u32 %8, %10, %8\n\t" INSN " %7, %8, %9, 0 offen\n\tv_add_u32 %8, %10, %8"       \
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]),  \
                   "+v"(voff)                                                                                            \
                 : "s"(rs), "n"(STRIDE))
    if constexpr (sizeof(K) == 8) KT_LD8("buffer_load_dwordx2");
    else KT_LD8("buffer_load_dword");
#undef KT_LD8
}
// (A store of more than 64 bits reads its data registers a cycle or two after it has issued: the compiler's hazard
// recognizer pads that for its own stores, not for inline assembly - the 16-byte stores of the write-combining kernels carry
// their own "s_nop 1".)
template <class K>
__device__ __forceinline__ void buf_store_async(K v, kt_i32x4 rs, uint32_t off) {
    if constexpr (sizeof(K) == 8) asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" ::"v"(v), "v"(off), "s"(rs));
    else asm volatile("buffer_store_dword %0, %1, %2, 0 offen" ::"v"(v), "v"(off), "s"(rs));
}
template <int N, class K, int PER>
__device__ __forceinline__ void buf_wait(K (&v)[PER]) {
    static_assert(PER % 8 == 0, "eight registers per asm statement");
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "n"(N));
#pragma unroll
    for (int u = 8; u < PER; u += 8)  // (volatile asm statements keep their order: these are behind the wait as well)
        asm volatile("" : "+v"(v[u]), "+v"(v[u + 1]), "+v"(v[u + 2]), "+v"(v[u + 3]), "+v"(v[u + 4]), "+v"(v[u + 5]), "+v"(v[u + 6]),
                          "+v"(v[u + 7]));
}

// the same wait for loads that went into a second set of registers: dst = src behind s_waitcnt vmcnt(N), the copy made
// inside the statement.  src is an input only: with the in-place form above on a second array the register allocator
// was free to tie the statement to dst's registers and copy src -> dst IN FRONT of the wait - registers whose loads had
// not landed (seen with 8 keys per thread: a few hundred wrong keys per run).  tools/check_inflight.py reads the
// kernel's assembly at build time: no instruction outside these statements may name a register between the asm load
// that targets it and the asm wait behind it.
template <int N, class K, int PER>
__device__ __forceinline__ void buf_take(K (&dst)[PER], K (&src)[PER]) {
    static_assert(PER % 8 == 0, "eight registers per asm statement");
#define KT_TAKE8(PRE, MOV, u)                                                                                              \
    asm volatile(PRE MOV " %0, %8\n\t" MOV " %1, %9\n\t" MOV " %2, %10\n\t" MOV " %3, %11\n\t" MOV " %4, %12\n\t" MOV             \
                 " %5, %13\n\t" MOV " %6, %14\n\t" MOV " %7, %15"                                                          \
                 : "=&v"(dst[u]), "=&v"(dst[u + 1]), "=&v"(dst[u + 2]), "=&v"(dst[u + 3]), "=&v"(dst[u + 4]),               \
                   "=&v"(dst[u + 5]), "=&v"(dst[u + 6]), "=&v"(dst[u + 7])                                                 \
                 : "v"(src[u]), "v"(src[u + 1]), "v"(src[u + 2]), "v"(src[u + 3]), "v"(src[u + 4]), "v"(src[u + 5]),         \
                   "v"(src[u + 6]), "v"(src[u + 7]), "n"(N))
    if constexpr (sizeof(K) == 8) KT_TAKE8("s_waitcnt vmcnt(%16)\n\t", "v_mov_b64", 0);
    else KT_TAKE8("s_waitcnt vmcnt(%16)\n\t", "v_mov_b32", 0);
#pragma unroll
    for (int u = 8; u < PER; u += 8) {  // (volatile asm statements keep their order: these are behind the wait as well)
        if constexpr (sizeof(K) == 8) KT_TAKE8("", "v_mov_b64", u);
        else KT_TAKE8("", "v_mov_b32", u);
    }
#undef KT_TAKE8
}

// wait for loads that are in flight into v[] and will never be used (a loop left early): the registers are INPUTS only, so
// nothing is copied anywhere - the point is that they are not handed to anything else while a load still targets them
template <class K, int PER>
__device__ __forceinline__ void buf_drain(K (&v)[PER]) {
    static_assert(PER % 8 == 0, "eight registers per asm statement");
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
#pragma unroll
    for (int u = 8; u < PER; u += 8)
        asm volatile("" ::"v"(v[u]), "v"(v[u + 1]), "v"(v[u + 2]), "v"(v[u + 3]), "v"(v[u + 4]), "v"(v[u + 5]), "v"(v[u + 6]), "v"(v[u + 7]));
}

using ktd::wave_incl_scan;

// block_excl_scan for B <= NT * PB with PB (counters per thread) known at compile time: the same contract - all NT threads
// call, two barriers, returns the total - in straight-line code: the wave's scan by DPP, the NT / 64 <= 16 wave totals
// scanned inside one DPP row.  (The general routine's per-thread loops with a run-time trip count cost the partition
// kernels a few hundred instructions and a dozen spilled registers per call.)
// (RND = 2^r - 1: every counter is rounded up to a multiple of 2^r first - runs that start on 16-byte boundaries, scatter1x;
// the counters are read through `get(i)`: scatter1y keeps them two to a word)
// (TAIL = false: without the closing barrier - the caller has one of its own before anybody reads another thread's out[])
template <int NT, int PB, uint32_t RND, bool TAIL = true, class Get>
__device__ __forceinline__ uint32_t block_excl_scan_f(Get get, uint32_t *out, uint32_t B, uint32_t *tmp) {
    static_assert(NT % 64 == 0 && NT / 64 <= 16, "wave totals fit one DPP row");
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t c[PB], sum = 0;
#pragma unroll
    for (int j = 0; j < PB; j++) {
        c[j] = tid * PB + j < B ? (get(tid * PB + j) + RND) & ~RND : 0u;
        sum += c[j];
    }
    const uint32_t inc = wave_incl_scan(sum);
    if (lane == 63) tmp[tid >> 6] = inc;
    ktd::lds_barrier();
    uint32_t w = lane < NT / 64 ? tmp[lane] : 0u;
    w += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x111, 0xf, 0xf, true);
    w += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x112, 0xf, 0xf, true);
    w += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x114, 0xf, 0xf, true);
    w += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x118, 0xf, 0xf, true);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)w, NT / 64 - 1);
    const uint32_t uw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t wbase = uw ? (uint32_t)__builtin_amdgcn_readlane((int)w, (int)uw - 1) : 0u;
    uint32_t run = wbase + inc - sum;
#pragma unroll
    for (int j = 0; j < PB; j++) {
        if (tid * PB + j < B) out[tid * PB + j] = run;
        run += c[j];
    }
    if constexpr (TAIL) ktd::lds_barrier();
    return total;
}
template <int NT, int PB, uint32_t RND = 0>
__device__ __forceinline__ uint32_t block_excl_scan_n(const uint32_t *cnt, uint32_t *out, uint32_t B, uint32_t *tmp) {
    return block_excl_scan_f<NT, PB, RND>([&](uint32_t i) { return cnt[i]; }, out, B, tmp);
}
// B <= 4 * NT (every caller: B <= 2048, NT >= 512), dispatched on a workgroup-uniform condition
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan_fast(const uint32_t *cnt, uint32_t *out, uint32_t B, uint32_t *tmp) {
    if (B <= (uint32_t)NT) return block_excl_scan_n<NT, 1>(cnt, out, B, tmp);
    if (B <= 2u * NT) return block_excl_scan_n<NT, 2>(cnt, out, B, tmp);
    return block_excl_scan_n<NT, 4>(cnt, out, B, tmp);
}

// ---- where the level-1 passes get their keys ---------------------------------------------------------------
// A source hands every workgroup "units" of up to 8192 canonical k-mers, 32 per thread in registers
// (bit j of `ok` = keys[j] is a k-mer).
struct ReadsSource {  // the k-mers of a read batch: unit = one 8192-base segment (kt_segment.hpp), segments [seg_lo, seg_hi)
    SegArgs a;
    uint64_t seg_lo, seg_hi;
    uint32_t n_parts, part;  // n_parts > 1: only the k-mers of hash partition `part` (out-of-core passes)
    __device__ uint64_t n_units() const { return seg_hi - seg_lo; }
    __device__ void collect(uint64_t g, SegShared &sm, uint64_t (&keys)[ktseg::PER_THREAD], uint32_t &ok) const {
        ktseg::collect_kmers(a, seg_lo + g, sm, keys, ok);
        if (n_parts > 1) {
#pragma unroll
            for (uint32_t j = 0; j < ktseg::PER_THREAD; j++)
                if (ktd::owner_of(keys[j], n_parts) != part) ok &= ~(1u << j);
        }
    }
    // the same k-mers a few at a time (scatter1x): open() stages the segment for the 256 threads of a group (t = index
    // in the group; whole-workgroup barriers inside), take<N>() hands out the thread's next N window starts
    struct Walk {
        ktseg::Window w;
        uint32_t at;
    };
    __device__ Walk open(uint64_t g, SegShared &sm, uint32_t t) const {
        ktseg::stage_segment(a, seg_lo + g, sm, t);
        return Walk{ktseg::Window(sm, t, a.k), 0u};
    }
    // open() with every read of the unit requested a unit ahead, from inline assembly (ktseg::prefetch_issue / _take)
    using Pre2 = ktseg::SegPrefetch2;
    using Taken = ktseg::SegTaken;
    __device__ uint64_t first_of(uint64_t g) const { return ktd::load_uniform(a.seg_first + ktd::uniform64(seg_lo + g)); }
    __device__ void prefetch_issue(Pre2 &pf, uint64_t g, uint64_t first_g, uint64_t g_next, uint32_t t, const void *safe) const {
        ktseg::prefetch_issue(pf, a, seg_lo + g, first_g, seg_lo + g_next, t, safe);
    }
    template <int NSTORE>
    __device__ void prefetch_take(Taken &tk, Pre2 &pf) const { ktseg::prefetch_take<NSTORE>(tk, pf); }
    __device__ Walk open_taken(uint64_t g, uint64_t first_g, SegShared &sm, uint32_t t, const Taken &tk) const {
        ktseg::stage_taken(a, seg_lo + g, first_g, sm, t, tk);
        return Walk{ktseg::Window(sm, t, a.k), 0u};
    }
    template <int N, class KR>  // KR = uint32_t when k <= 16: half the registers
    __device__ void take(Walk &wk, uint32_t, KR (&keys)[N], uint32_t &ok) const {
        ok = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            const uint64_t m = wk.w.f < wk.w.r ? wk.w.f : wk.w.r;
            keys[j] = (KR)m;
            bool good = wk.w.ok(wk.at + j);
            if (n_parts > 1) good = good && ktd::owner_of(m, n_parts) == part;
            ok |= (good ? 1u : 0u) << j;
            wk.w.step();
        }
        wk.at += N;
    }
    template <class Sink>
    __device__ void for_each(uint64_t g, SegShared &sm, Sink &&sink) const {  // rolling walk: few registers
        ktseg::for_each_kmer(a, seg_lo + g, sm, [&](uint64_t f, uint64_t r, uint64_t) {
            const uint64_t m = f < r ? f : r;
            if (n_parts <= 1 || ktd::owner_of(m, n_parts) == part) sink(m);
        });
    }
};
// The same k-mers from reads that a pass of their own has PACKED (pack_segments_kernel: what stage_segment makes of a
// segment - 32 bases as a 64-bit word of codes, the invalid-base mask, the read-start mask: 16 bytes per 32 bases - written
// out instead of kept in LDS): level 1 then stages nothing - a thread's window words are its item and the next one, 32
// contiguous bytes - and has no barrier of its own per unit.
struct PackedSource {
    const uint4 *items;  // item i = bases [32 i, 32 i + 32): {codes lo, codes hi, invalid mask, read-start mask}; one terminator behind the last
    uint64_t seg_lo, seg_hi;
    uint32_t k;
    __device__ uint64_t n_units() const { return seg_hi - seg_lo; }
    struct Walk {
        ktseg::Window w;
        uint32_t at;
    };
    static __device__ __forceinline__ Walk walk_of(const uint64_t (&a)[4], uint32_t k) {
        // a[0] = codes of item t, a[1] = {inv, bnd} of item t, a[2] / a[3] the same of item t + 1
        const uint64_t iv = (a[1] & 0xFFFFFFFFull) | (a[3] << 32), bd = (a[1] >> 32) | (a[3] & 0xFFFFFFFF00000000ull);
        return Walk{ktseg::Window(a[0], a[2], iv, bd, k), 0u};
    }
    __device__ Walk open(uint64_t g, SegShared &, uint32_t t) const {
        const uint64_t *p = reinterpret_cast<const uint64_t *>(items + (seg_lo + g) * BLOCK + t);
        const uint64_t a[4] = {p[0], p[1], p[2], p[3]};
        return walk_of(a, k);
    }
    struct Pre2 {
        uint64_t a[4], o0;  // in flight until prefetch_take
    };
    struct Taken {
        uint64_t a[4], first_next;
    };
    __device__ uint64_t first_of(uint64_t) const { return 0; }
    __device__ void prefetch_issue(Pre2 &pf, uint64_t g, uint64_t, uint64_t, uint32_t t, const void *safe) const {
        const void *p0 = items + (ktd::uniform64(seg_lo + g) * BLOCK + t);
        asm volatile("global_load_dwordx2 %0, %5, off\n\tglobal_load_dwordx2 %1, %5, off offset:8\n\t"
                     "global_load_dwordx2 %2, %5, off offset:16\n\tglobal_load_dwordx2 %3, %5, off offset:24\n\t"
                     "global_load_dwordx2 %4, %6, off"
                     : "=&v"(pf.a[0]), "=&v"(pf.a[1]), "=&v"(pf.a[2]), "=&v"(pf.a[3]), "=&v"(pf.o0) : "v"(p0), "v"(safe) : "memory");
    }
    template <int NSTORE>
    __device__ void prefetch_take(Taken &tk, Pre2 &pf) const {
        uint64_t o0;
        asm volatile("s_waitcnt vmcnt(%10)\n\tv_mov_b64 %0, %5\n\tv_mov_b64 %1, %6\n\tv_mov_b64 %2, %7\n\tv_mov_b64 %3, %8\n\tv_mov_b64 %4, %9"
                     : "=&v"(tk.a[0]), "=&v"(tk.a[1]), "=&v"(tk.a[2]), "=&v"(tk.a[3]), "=&v"(o0)
                     : "v"(pf.a[0]), "v"(pf.a[1]), "v"(pf.a[2]), "v"(pf.a[3]), "v"(pf.o0), "n"(NSTORE)
                     : "memory");
        tk.first_next = 0;
    }
    __device__ Walk open_taken(uint64_t, uint64_t, SegShared &, uint32_t, const Taken &tk) const { return walk_of(tk.a, k); }
    template <int N, class KR>
    __device__ void take(Walk &wk, uint32_t, KR (&keys)[N], uint32_t &ok) const {
        ok = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            keys[j] = (KR)(wk.w.f < wk.w.r ? wk.w.f : wk.w.r);
            ok |= (wk.w.ok(wk.at + j) ? 1u : 0u) << j;
            wk.w.step();
        }
        wk.at += N;
    }
    __device__ void collect(uint64_t g, SegShared &sm, uint64_t (&keys)[ktseg::PER_THREAD], uint32_t &ok) const {
        Walk wk = open(g, sm, threadIdx.x);
        take<(int)ktseg::PER_THREAD, uint64_t>(wk, threadIdx.x, keys, ok);
    }
    template <class Sink>
    __device__ void for_each(uint64_t g, SegShared &sm, Sink &&sink) const {
        Walk wk = open(g, sm, threadIdx.x);
#pragma unroll 4
        for (uint32_t j = 0; j < ktseg::PER_THREAD; j++) {
            if (wk.w.ok(j)) sink(wk.w.f < wk.w.r ? wk.w.f : wk.w.r);
            wk.w.step();
        }
    }
};

// what stage_segment makes of the segments [seg_lo, seg_hi), written out: item (g, t) at items[g * 256 + t].  A thread's codes
// and invalid-base mask come straight from the 32 bytes it has read; only the read starts cross threads (the thread that reads
// read r's offset sets a bit in another thread's item): one LDS word per item, two sets of them - a thread clears its word of
// the set the NEXT segment will use in front of this segment's barrier - so a segment costs one barrier.  The reads of a
// segment are asked for while the one before it is packed (ktseg::request_ahead).
__global__ __launch_bounds__(BLOCK) void pack_segments_kernel(SegArgs a, uint64_t seg_lo, uint64_t seg_hi, uint4 *__restrict__ items) {
    __shared__ uint32_t bnd2[2][BLOCK];
    const uint32_t tid = threadIdx.x;
    const uint64_t g_first = seg_lo + blockIdx.x;
    const uint64_t total = ktd::load_uniform(a.offsets + a.n_reads);
    ktseg::SegAhead ah{};
    if (g_first < seg_hi)
        ah = ktseg::request_ahead(a, g_first, ktd::load_uniform(a.seg_first + ktd::uniform64(g_first)), g_first + gridDim.x, tid);
    bnd2[0][tid] = 0;
    ktd::lds_barrier();
    uint32_t par = 0;
    for (uint64_t g = g_first; g < seg_hi; g += gridDim.x, par ^= 1u) {
        const uint64_t B0 = g * ktseg::SEG, first_g = ah.first, first_next = ah.first_next, o0 = ah.o0;
        uint64_t w;
        uint32_t iv;
        ktseg::encode_item(a, total, B0 + 32ull * tid, ah.d0, ah.whole0, w, iv);
        if (g + gridDim.x < seg_hi) ah = ktseg::request_ahead(a, g + gridDim.x, first_next, g + 2ull * gridDim.x, tid);
        uint32_t *const bnd = bnd2[par];
        bnd2[par ^ 1u][tid] = 0;  // (last read a barrier ago, by this thread; next written behind the barrier below)
        {
            const uint64_t lim = B0 + ktseg::SEG, r0 = first_g + tid;
            if (r0 < a.n_reads && o0 < lim) {
                const uint32_t rel = (uint32_t)(o0 - B0);
                atomicOr(&bnd[rel >> 5], 1u << (rel & 31u));
                for (uint64_t r = r0 + BLOCK; r < a.n_reads; r += BLOCK) {  // more than 256 read starts in a segment
                    const uint64_t o = a.offsets[r];
                    if (o >= lim) break;
                    const uint32_t rel2 = (uint32_t)(o - B0);
                    atomicOr(&bnd[rel2 >> 5], 1u << (rel2 & 31u));
                }
            }
        }
        ktd::lds_barrier();
        typedef uint32_t raw4 __attribute__((ext_vector_type(4)));
        const raw4 v = {(uint32_t)w, (uint32_t)(w >> 32), iv, bnd[tid]};
        *reinterpret_cast<raw4 *>(items + g * BLOCK + tid) = v;
    }
    if (blockIdx.x == 0 && tid == 0) items[a.n_seg * BLOCK] = make_uint4(0u, 0u, 0xFFFFFFFFu, 0u);  // behind the last item: nothing
}

struct KeysSource {  // canonical k-mers that are already an array: unit = 8192 keys
    const uint64_t *keys;
    uint64_t n;
    __device__ uint64_t count() const { return n; }
    __device__ uint64_t n_units() const { return (count() + ktseg::SEG - 1) / ktseg::SEG; }
    __device__ void collect(uint64_t g, SegShared &, uint64_t (&out)[ktseg::PER_THREAD], uint32_t &ok) const {
        const uint64_t cnt = count();
        ok = 0;
#pragma unroll
        for (uint32_t j = 0; j < ktseg::PER_THREAD; j++) {  // consecutive lanes read consecutive keys
            const uint64_t i = g * ktseg::SEG + (uint64_t)j * BLOCK + threadIdx.x;
            const uint64_t key = i < cnt ? keys[i] : KT_EMPTY_KEY;
            out[j] = key;
            ok |= (key != KT_EMPTY_KEY ? 1u : 0u) << j;
        }
    }
    struct Walk {
        uint64_t base, cnt;
        uint32_t at;
    };
    __device__ Walk open(uint64_t g, SegShared &, uint32_t) const { return Walk{g * ktseg::SEG, count(), 0u}; }
    struct Pre2 {};
    struct Taken { uint64_t first_next; };
    __device__ uint64_t first_of(uint64_t) const { return 0; }
    __device__ void prefetch_issue(Pre2 &, uint64_t, uint64_t, uint64_t, uint32_t, const void *) const {}
    template <int NSTORE>
    __device__ void prefetch_take(Taken &tk, Pre2 &) const { tk.first_next = 0; }
    __device__ Walk open_taken(uint64_t g, uint64_t, SegShared &sm, uint32_t t, const Taken &) const { return open(g, sm, t); }
    template <int N, class KR>
    __device__ void take(Walk &wk, uint32_t t, KR (&out)[N], uint32_t &ok) const {
        ok = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {  // consecutive lanes read consecutive keys
            const uint64_t i = wk.base + (uint64_t)(wk.at + j) * BLOCK + t;
            const uint64_t key = i < wk.cnt ? keys[i] : KT_EMPTY_KEY;
            out[j] = (KR)key;
            ok |= (key != KT_EMPTY_KEY ? 1u : 0u) << j;
        }
        wk.at += N;
    }
    template <class Sink>
    __device__ void for_each(uint64_t g, SegShared &, Sink &&sink) const {
        const uint64_t cnt = count();
#pragma unroll 8
        for (uint32_t j = 0; j < ktseg::PER_THREAD; j++) {
            const uint64_t i = g * ktseg::SEG + (uint64_t)j * BLOCK + threadIdx.x;
            const uint64_t key = i < cnt ? keys[i] : KT_EMPTY_KEY;
            if (key != KT_EMPTY_KEY) sink(key);
        }
    }
};

// The k-mers of RECORDS (kt_superkmer.hpp: at most 8 consecutive k-mers of a read as 8 + k - 1 bases at 2 bits - what the
// ranks of the sharded counter send each other): unit = one block of 1024 records = 4 per thread, i.e. the same 32 window
// starts per thread and unit as a segment of reads, a record per 8 of them.  Nothing is staged through LDS: a thread's
// four records are 32 + 8 contiguous bytes of the block.
struct RecordSource {
    const ktsk::RecRun *runs;  // (device) the stretches of whole blocks this source covers, in order
    const uint64_t *cum;       // (device) [n_runs + 1] blocks in the stretches before run j
    uint32_t n_runs, k;
    uint64_t n_blocks;
    __device__ uint64_t n_units() const { return n_blocks; }
    // block g of the source and the records it holds (g: the same in every lane of the wave)
    __device__ void locate(uint64_t g, const uint64_t *&blk, uint32_t &n_in) const {
        uint32_t lo = 0, hi = n_runs;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (ktd::load_uniform(cum + mid) <= g) lo = mid;
            else hi = mid;
        }
        const uint64_t b = g - ktd::load_uniform(cum + lo);
        const uint64_t *const base = ktd::load_uniform(&runs[lo].blocks);
        const uint64_t left = ktd::load_uniform(&runs[lo].n_rec) - b * ktsk::BLOCK_RECS;
        blk = base + b * ktsk::BLOCK_WORDS;
        n_in = left < ktsk::BLOCK_RECS ? (uint32_t)left : ktsk::BLOCK_RECS;
    }
    struct Walk {
        uint64_t a[4], bw;  // the thread's four records: a, and the four b side by side (record c in bits 16c .. 16c + 15)
        uint32_t at;
    };
    // records at and beyond n_in (the unfilled tail of a region's last block) hold nothing
    static __device__ __forceinline__ uint64_t clip(uint64_t bw, uint32_t t, uint32_t n_in) {
        const uint32_t have = n_in > 4u * t ? n_in - 4u * t : 0u;
        return have >= 4u ? bw : bw & ((1ull << (16u * have)) - 1ull);
    }
    __device__ Walk open(uint64_t g, SegShared &, uint32_t t) const {
        const uint64_t *blk;
        uint32_t n_in;
        locate(ktd::uniform64(g), blk, n_in);
        Walk wk;
#pragma unroll
        for (int c = 0; c < 4; c++) wk.a[c] = blk[4 * t + c];
        wk.bw = clip(blk[ktsk::BLOCK_RECS + t], t, n_in);
        wk.at = 0;
        return wk;
    }
    // the same with the unit's reads requested a unit ahead, from inline assembly (see ktseg::prefetch_issue: five
    // loads, the same shape as a segment's)
    struct Pre2 {
        uint64_t a[4], bw;  // in flight until prefetch_take
        uint32_t n_in;
    };
    struct Taken {
        uint64_t a[4], bw, first_next;
        uint32_t n_in;
    };
    __device__ uint64_t first_of(uint64_t) const { return 0; }
    __device__ void prefetch_issue(Pre2 &pf, uint64_t g, uint64_t, uint64_t, uint32_t t, const void *) const {
        const uint64_t *blk;
        locate(ktd::uniform64(g), blk, pf.n_in);
        const void *p0 = blk + 4 * t, *pb = blk + ktsk::BLOCK_RECS + t;
        asm volatile("global_load_dwordx2 %0, %5, off\n\tglobal_load_dwordx2 %1, %5, off offset:8\n\t"
                     "global_load_dwordx2 %2, %5, off offset:16\n\tglobal_load_dwordx2 %3, %5, off offset:24\n\t"
                     "global_load_dwordx2 %4, %6, off"
                     : "=&v"(pf.a[0]), "=&v"(pf.a[1]), "=&v"(pf.a[2]), "=&v"(pf.a[3]), "=&v"(pf.bw) : "v"(p0), "v"(pb) : "memory");
    }
    template <int NSTORE>
    __device__ void prefetch_take(Taken &tk, Pre2 &pf) const {
        // (the in-flight registers are inputs only: see ktseg::prefetch_take)
        asm volatile("s_waitcnt vmcnt(%10)\n\tv_mov_b64 %0, %5\n\tv_mov_b64 %1, %6\n\tv_mov_b64 %2, %7\n\tv_mov_b64 %3, %8\n\tv_mov_b64 %4, %9"
                     : "=&v"(tk.a[0]), "=&v"(tk.a[1]), "=&v"(tk.a[2]), "=&v"(tk.a[3]), "=&v"(tk.bw)
                     : "v"(pf.a[0]), "v"(pf.a[1]), "v"(pf.a[2]), "v"(pf.a[3]), "v"(pf.bw), "n"(NSTORE)
                     : "memory");
        tk.n_in = pf.n_in;
        tk.first_next = 0;
    }
    __device__ Walk open_taken(uint64_t, uint64_t, SegShared &, uint32_t t, const Taken &tk) const {
        Walk wk;
#pragma unroll
        for (int c = 0; c < 4; c++) wk.a[c] = tk.a[c];
        wk.bw = clip(tk.bw, t, tk.n_in);
        wk.at = 0;
        return wk;
    }
    // the thread's next N window starts: N / 8 records, eight windows each (bit j of ok: the record has a k-mer there)
    template <int N, class KR>
    __device__ void take(Walk &wk, uint32_t, KR (&keys)[N], uint32_t &ok) const {
        static_assert(N % (int)ktsk::REC_KMERS == 0, "whole records per round");
        ok = 0;
#pragma unroll
        for (int c = 0; c < N / 8; c++) {
            const uint32_t rec = wk.at / 8u + (uint32_t)c;
            const uint64_t a = rec == 0 ? wk.a[0] : rec == 1 ? wk.a[1] : rec == 2 ? wk.a[2] : wk.a[3];
            const uint32_t b = (uint32_t)(wk.bw >> (16u * rec)) & 0xFFFFu;
            const uint32_t n = b & 15u;
            ktseg::Window w(a, (uint64_t)(b & 0xFFF0u) << 48, (1u << n) - 1u, k);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                keys[c * 8 + j] = (KR)(w.f < w.r ? w.f : w.r);
                w.step();
            }
            ok |= ((1u << n) - 1u) << (c * 8);
        }
        wk.at += N;
    }
    __device__ void collect(uint64_t g, SegShared &sm, uint64_t (&keys)[ktseg::PER_THREAD], uint32_t &ok) const {
        Walk wk = open(g, sm, threadIdx.x);
        take<(int)ktseg::PER_THREAD, uint64_t>(wk, threadIdx.x, keys, ok);
    }
    template <class Sink>
    __device__ void for_each(uint64_t g, SegShared &sm, Sink &&sink) const {
        Walk wk = open(g, sm, threadIdx.x);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t b = (uint32_t)(wk.bw >> (16 * c)) & 0xFFFFu;
            const uint32_t n = b & 15u;
            ktseg::Window w(wk.a[c], (uint64_t)(b & 0xFFF0u) << 48, (1u << n) - 1u, k);
            for (uint32_t j = 0; j < n; j++) {
                sink(w.f < w.r ? w.f : w.r);
                w.step();
            }
        }
    }
};

// ---- hist1 --------------------------------------------------------------------------------------
template <class Source>
__global__ __launch_bounds__(BLOCK) void hist1_kernel(Source src, Plan p, uint32_t *__restrict__ H) {
    __shared__ SegShared sm;
    __shared__ uint32_t cnt[MAX_B];
    for (uint32_t i = threadIdx.x; i < p.B1; i += BLOCK) cnt[i] = 0;
    ktd::lds_barrier();
    const uint64_t n_units = src.n_units();
    for (uint64_t g = blockIdx.x; g < n_units; g += gridDim.x) {
        src.for_each(g, sm, [&](uint64_t key) { atomicAdd(&cnt[digit1(key, p)], 1u); });
    }
    ktd::lds_barrier();
    for (uint32_t i = threadIdx.x; i < p.B1; i += BLOCK) H[(uint64_t)blockIdx.x * p.B1 + i] += cnt[i];  // (a job may have several sources)
}

// records through the probing path (a batch too small for the partition passes, or small against a table that holds data)
__global__ __launch_bounds__(BLOCK) void count_records_kernel(RecordSource src, TableRef t, uint64_t *__restrict__ distinct) {
    __shared__ SegShared sm;  // (not used by this source)
    uint32_t fresh = 0;
    const uint64_t n_units = src.n_units();
    for (uint64_t g = blockIdx.x; g < n_units; g += gridDim.x) {
        src.for_each(g, sm, [&](uint64_t key) {
            const uint32_t st = kttab::table_add(t, key, 1u);
            if (st == 0u) atomicOr(t.flags, 1u);
            fresh += st == 2u;
        });
    }
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_down(fresh, o, 64);
    if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(reinterpret_cast<unsigned long long *>(distinct), (unsigned long long)fresh);
}

// ---- scan1: O[g][d] = (k-mers in buckets < d) + (k-mers of workgroups < g in bucket d) ------------
__global__ __launch_bounds__(1024) void scan1_kernel(const uint32_t *__restrict__ H, Plan p, uint64_t *__restrict__ O,
                                                     uint64_t *__restrict__ bstart) {
    __shared__ uint64_t tot[MAX_B];
    for (uint32_t d = threadIdx.x; d < p.B1; d += blockDim.x) {
        uint64_t run = 0;
        for (uint32_t g = 0; g < p.G; g++) {
            O[(uint64_t)g * p.B1 + d] = run;
            run += H[(uint64_t)g * p.B1 + d];
        }
        tot[d] = run;
    }
    ktd::lds_barrier();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (uint32_t d = 0; d < p.B1; d++) {
            const uint64_t c = tot[d];
            tot[d] = run;
            bstart[d] = run;
            run += c;
        }
        bstart[p.B1] = run;
    }
    ktd::lds_barrier();
    for (uint32_t d = threadIdx.x; d < p.B1; d += blockDim.x) {
        const uint64_t base = tot[d];
        for (uint32_t g = 0; g < p.G; g++) O[(uint64_t)g * p.B1 + d] += base;
    }
}

// ---- scatter1 ----------------------------------------------------------------------------------------
// keys sorted at a time: half a unit of 64-bit keys (two rounds), a whole unit of 32-bit keys
template <class K>
constexpr uint32_t round_keys() { return sizeof(K) == 8 ? ktseg::SEG / 2 : ktseg::SEG; }

constexpr uint32_t MAX_B1 = 1024;  // level 1 never uses more than 10 bits (bulk_build_from)

template <class K>
struct Scatter1Shared {  // 65 / 73 KB: two workgroups per CU (an 85 KB layout dropped to one and cost 10 %)
    SegShared seg;
    K sorted[round_keys<K>()];
    uint16_t sdig[round_keys<K>()];  // level-1 digit of sorted[i]: saves hashing the key a third time
    uint64_t cursor[MAX_B1];
    uint32_t cnt[MAX_B1];
    uint32_t start[MAX_B1];
    uint32_t fill[MAX_B1];
    uint32_t tmp[BLOCK];
};
static_assert(sizeof(Scatter1Shared<uint64_t>) <= 80 * 1024 && sizeof(Scatter1Shared<uint32_t>) <= 80 * 1024,
              "two scatter1 workgroups per CU");

template <class Source, class K>
__global__ __launch_bounds__(BLOCK) void scatter1_kernel(Source src, Plan p, uint64_t *__restrict__ O,
                                                         K *__restrict__ keys1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Scatter1Shared<K> &sm = *reinterpret_cast<Scatter1Shared<K> *>(smem_raw);
    constexpr int ROUNDS = ktseg::SEG / round_keys<K>(), PERR = ktseg::PER_THREAD / ROUNDS;
    for (uint32_t i = threadIdx.x; i < p.B1; i += BLOCK) sm.cursor[i] = O[(uint64_t)blockIdx.x * p.B1 + i];
    const uint64_t n_units = src.n_units();
    for (uint64_t g = blockIdx.x; g < n_units; g += gridDim.x) {
        // one front-end pass: the thread's 32 canonical k-mers stay in registers
        uint64_t keys[ktseg::PER_THREAD];
        uint32_t ok;
        src.collect(g, sm.seg, keys, ok);
#pragma unroll
        for (int half = 0; half < ROUNDS; half++) {
            for (uint32_t i = threadIdx.x; i < p.B1; i += BLOCK) sm.cnt[i] = 0;
            ktd::lds_barrier();
#pragma unroll
            for (int j = 0; j < PERR; j++)
            {  // (the keys take their stored form here, where the first digit is needed)
                keys[half * PERR + j] = (uint64_t)to_stored<K>(keys[half * PERR + j]);
                if ((ok >> (half * PERR + j)) & 1u)
                    atomicAdd(&sm.cnt[digit1h(hash_of_stored<K>((K)keys[half * PERR + j], p), p)], 1u);
            }
            ktd::lds_barrier();
            const uint32_t nk = block_excl_scan(sm.cnt, sm.start, p.B1, sm.tmp);
            for (uint32_t i = threadIdx.x; i < p.B1; i += BLOCK) sm.fill[i] = sm.start[i];
            ktd::lds_barrier();
#pragma unroll
            for (int j = 0; j < PERR; j++) {
                if ((ok >> (half * PERR + j)) & 1u) {
                    const uint64_t m = keys[half * PERR + j];
                    const uint32_t d = digit1h(hash_of_stored<K>((K)m, p), p);
                    const uint32_t pos = atomicAdd(&sm.fill[d], 1u);
                    sm.sorted[pos] = (K)m;
                    sm.sdig[pos] = (uint16_t)d;
                }
            }
            ktd::lds_barrier();
            // runs of equal d1 are contiguous in `sorted`: consecutive lanes -> consecutive addresses
            for (uint32_t i = threadIdx.x; i < nk; i += BLOCK) {
                const uint32_t d = sm.sdig[i];
                keys1[sm.cursor[d] + (i - sm.start[d])] = sm.sorted[i];
            }
            ktd::lds_barrier();
            for (uint32_t i = threadIdx.x; i < p.B1; i += BLOCK) sm.cursor[i] += sm.cnt[i];
            ktd::lds_barrier();
        }
    }
    // the next source of the job carries on where this one stopped
    for (uint32_t i = threadIdx.x; i < p.B1; i += BLOCK) O[(uint64_t)blockIdx.x * p.B1 + i] = sm.cursor[i];
}

// ---- level 1, the kernel the bulk path runs: scatter1x --------------------------------------------------------------------
// The exact path above needs hist1, a whole front-end pass whose only purpose is to make scatter1's output offsets
// exact.  The bulk path drops it: every level-1 bucket owns a fixed REGION of cap1 keys and the workgroups append to it; a
// bucket that runs out of room (a batch dominated by a few k-mers) raises *ovf and the build is redone through hist1 /
// scan1 / scatter1.  Rounds 2-4 appended through workgroup-private aligned pages (scatter1p: one 256-thread workgroup per
// unit, 32-byte runs; scatter1w: 1024 threads, four segments at a time, 128-byte runs, 16 K keys sorted per round) - the
// shape below keeps scatter1w's rounds and replaces its pages.
#if KT_ABLATION
__device__ unsigned long long kt_dbg_phase[16];  // (timing builds: cycles of workgroup thread 0 per phase: [0, 8) build_kernel, [8, 16) scatter1x_kernel)
#define KT_PH(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); phs[i] += (uint32_t)(t_ - tph); tph = t_; } while (0)
#else
#define KT_PH(i) do { } while (0)
#endif

#ifndef KT_BUILD_NT
#define KT_BUILD_NT 1  // the table image is written once, whole lines at a time: non-temporal stores (k=31 -5 %).  The
#endif                 // partition passes' short key runs need the L2 to merge them: there the same hint costs 50 %.
#ifndef KT_WIDE_PER64
#define KT_WIDE_PER64 16
#endif
template <class K>
constexpr int wide_per() { return sizeof(K) == 8 ? KT_WIDE_PER64 : 32; }  // keys per thread and round

// ---- scatter1x: level 1 with ONE CURSOR PER BUCKET AND XCD (round 5) --------------------------------------------------
// scatter1w's appends go through workgroup-private pages: the open line of a bucket is completed by the same workgroup a
// round (~18 us) later, and 32 workgroups x 1024 buckets x 128 bytes of open lines are the whole 4 MB L2 of an XCD - the
// lines leave the L2 half written: 31 GB of write requests for 24 GB of keys, at the 1.5-1.9 TB/s of partial granules
// (tools/ubench/scatter_runs.hip), a third of the kernel's time and nothing else running meanwhile.  Here the workgroups
// that share an L2 share the cursor: a workgroup takes its run's place with one returning atomic per (bucket, round) on
// the cursor set of ITS XCD (the hardware's XCC_ID), so an XCD has B1 open lines (128 KB), a line is completed by its
// neighbours within a microsecond, and the L2 evicts whole lines - tools/ubench/xcd_append.hip: 128-byte runs 1.85 ->
// 4.6 TB/s, 64-byte runs 1.36 -> 4.0, the atomics themselves 0.4 ms per 8 GB (agent scope: correct whatever the
// placement; the XCC_ID only decides which lines a cache shares).  Layout: a bucket's region is cut into lines of LK keys
// dealt round-robin to the cursor sets - key q of set x lies at ((q / LK) * NX + x) * LK + q % LK - so a line is written
// by one XCD only, the region stays ONE array for level 2, and what differs
// between the sets' fills is padded with the empty key by xcd_tails_kernel (a few hundred keys per bucket and set).  No
// pages, no state carried between launches but the cursors themselves, no split runs in the copy-out.
// With the stores cheap, what is left is the phases' serial chain - so the workgroup is HALF the size (T = 512: two per CU,
// 8 K 64-bit or 16 K 32-bit keys per round): the two run out of phase by themselves, one's front end under the other's
// copy-out (the 64-byte runs that would have cost the private pages dearly cost the shared cursors nothing).
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
__global__ void xcc_census_kernel(uint32_t *mask) {
    if (threadIdx.x == 0) atomicOr(mask, 1u << xcc_id());
}

#ifndef KT_S1X_T
#define KT_S1X_T 1024
#endif
// keys of one 16-byte store (2 / 4): every bucket's run of a round starts on a multiple of it - in the sort buffer and in the
// bucket's stream - and is padded to one with the empty key, so that a lane copies a GROUP with one 16-byte LDS read and one
// 16-byte store.  Why: 8-byte stores per lane are bound by their ISSUE - ~7 bytes per clock and CU whatever the HBM could
// take (MI355X_MICROARCH.md, the store-tail row; the same 7.3 here, in tools/ubench/xcd_append.hip and in the kernel's own
// copy-out) - and a 16-byte store is issued in the time of an 8-byte one.  (4-byte keys stored one per lane were half as
// fast again.)  The pads cost a few per cent of the keys' bytes (half a group per run), which level 2 skips as it
// skips the regions' tails.
template <class K>
constexpr uint32_t group_keys() { return 16 / sizeof(K); }
template <class K, int T>
constexpr uint32_t s1x_slots() { return (uint32_t)T * wide_per<K>() + MAX_B1 * (group_keys<K>() - 1); }  // keys + pads of a round
template <class K, int T>
struct Scatter1XShared {
    union {  // (the staged segments are dead once every thread has loaded its window words: open())
        SegShared seg[T / BLOCK];
        K sorted[s1x_slots<K, T>()];
    };
    uint32_t cnt[MAX_B1];
    uint32_t start[MAX_B1];
    uint32_t delta[MAX_B1];  // (the run's first place in its set's stream of the bucket) - (the run's start in sorted[])
    uint32_t tmp[16];
    uint32_t ovf;
};
static_assert(sizeof(Scatter1XShared<uint64_t, KT_S1X_T>) <= 160 * 1024 && sizeof(Scatter1XShared<uint32_t, KT_S1X_T>) <= 160 * 1024, "LDS of a CU");

// where key q of cursor set x lies in its bucket's region
template <class K>
__device__ __forceinline__ uint64_t xcd_place(uint64_t q, uint32_t x, uint32_t nxs) {
    constexpr uint32_t LSH = sizeof(K) == 8 ? 4 : 5;  // 16 / 32 keys per 128-byte line
    return ((((q >> LSH) << nxs) | x) << LSH) | (q & ((1u << LSH) - 1u));
}

template <class Source, class K, int T>
__global__ __launch_bounds__(T, 4) void scatter1x_kernel(Source src, Plan p, unsigned long long *__restrict__ xcur,
                                                      uint32_t *__restrict__ ovf, K *__restrict__ keys1,
                                                      unsigned long long *__restrict__ dump) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Scatter1XShared<K, T> &sm = *reinterpret_cast<Scatter1XShared<K, T> *>(smem_raw);
    constexpr int PER = wide_per<K>(), NQ = ktseg::PER_THREAD / PER, GROUPS = T / BLOCK, PB = MAX_B1 / T;
    constexpr uint32_t GK = group_keys<K>(), GSH = GK == 2 ? 1 : 2;
    // 16-byte stores per thread and round: the round's keys + what the pads can add, in groups, over the threads
    constexpr int NST = (int)((s1x_slots<K, T>() / GK + T - 1) / T);
    constexpr K EMPTY = empty_of<K>();
    typedef uint32_t raw4 __attribute__((ext_vector_type(4)));
    const uint32_t tid = threadIdx.x, grp = tid / BLOCK, t = tid % BLOCK;
    const uint32_t xset = (p.xmap >> (4u * xcc_id())) & 7u;
    unsigned long long *const mycur = xcur + (size_t)xset * p.B1;
#pragma unroll
    for (int j = 0; j < PB; j++) sm.cnt[tid * PB + j] = 0;  // (every round leaves cnt zeroed for the next one)
    if (tid == 0) sm.ovf = 0;
    const uint64_t n_units = src.n_units();
    bool stop = false;
    const uint64_t stride = (uint64_t)gridDim.x * GROUPS;
    auto unit_of = [&](uint64_t g0) { return g0 + grp < n_units ? g0 + grp : n_units - 1; };
    // Every global read of a unit is requested while the unit before it is still being counted - its bases, the thread's
    // first read start, and seg_first of the unit after (ktseg::prefetch_issue) - BEFORE the stores of that unit's last
    // copy-out, and taken behind them with s_waitcnt vmcnt(NST): the reads have landed, the NST stores stay in flight.
    // For that the copy-out issues exactly NST stores on every path - a group past the end of the round, or past its
    // region's room, is stored to a dump line instead of being skipped.
    typename Source::Pre2 pre;
    typename Source::Taken tk{};
    uint64_t first_cur = 0;
    {
        const uint64_t gfirst = (uint64_t)blockIdx.x * GROUPS;
        if (gfirst < n_units) {
            first_cur = src.first_of(unit_of(gfirst));
            src.prefetch_issue(pre, unit_of(gfirst), first_cur, unit_of(gfirst + stride), t, dump);
            src.template prefetch_take<0>(tk, pre);
        }
    }
#if KT_ABLATION
    unsigned long long tph = __builtin_readcyclecounter();
    uint32_t phs[8] = {};
#endif
    for (uint64_t g0 = (uint64_t)blockIdx.x * GROUPS; g0 < n_units && !stop; g0 += stride) {
        const bool valid = g0 + grp < n_units;  // (a group past the end walks the last unit and keeps nothing)
        auto wk = src.open_taken(unit_of(g0), first_cur, sm.seg[grp], t, tk);
        const uint64_t first_next = tk.first_next;
        KT_PH(0);
#pragma unroll
        for (int q = 0; q < NQ; q++) {  // (no early exit in here: the loop must stay unrolled)
            // (the thread index, opaque per round: nothing derived from it - bucket addresses, the cursors' addresses, the
            // copy-out's indices - is carried across the loop in registers the kernel does not have)
            uint32_t tl = tid;
            asm volatile("" : "+v"(tl));
            K keys[PER];
            uint32_t ok;
            src.template take<PER, K>(wk, tl % BLOCK, keys, ok);
            if (!valid) ok = 0;
#pragma unroll
            for (int j = 0; j < PER; j++) {  // (the keys take their stored form here, where the first digit is needed)
                keys[j] = to_stored<K>((uint64_t)keys[j]);
                if ((ok >> j) & 1u) atomicAdd(&sm.cnt[digit1h(hash_of_stored<K>(keys[j], p), p)], 1u);
            }
            KT_PH(1);
            ktd::lds_barrier();
            KT_PH(2);
            // the runs of the round in sorted[], every one starting on a group boundary: nk = the slots they take, pads included
            const uint32_t nk = block_excl_scan_n<T, PB, GK - 1>(sm.cnt, sm.start, p.B1, sm.tmp);
            // every bucket's run takes its place in the XCD's stream of the bucket: one returning atomic, whose answer is
            // only looked at behind the placement pass (which hides the round trip)
            unsigned long long got[PB];
            uint32_t rc[PB], rs[PB];
#pragma unroll
            for (int j = 0; j < PB; j++) {
                // (no branch around the atomic - a bucket beyond B1, or without keys this round, adds 0 to a cursor of the
                // set: under a condition the compiler waits for the answer where the branches meet, i.e. at once, and the
                // round trip it was meant to hide behind the placement pass is paid in full - measured: a fifth of the kernel)
                const uint32_t d = tl * PB + j, dc = d & (p.B1 - 1u);
                rc[j] = d < p.B1 ? sm.cnt[dc] : 0u;
                rs[j] = sm.start[dc];
                if (d < p.B1) sm.cnt[d] = 0;  // for the next round's count
                got[j] = __hip_atomic_fetch_add(&mycur[dc], (unsigned long long)((rc[j] + GK - 1u) & ~(GK - 1u)), __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
            }
            KT_PH(3);
            ktd::lds_barrier();  // start[] was read above; the placement pass uses it as its cursors
            KT_PH(4);
#pragma unroll
            for (int j = 0; j < PER; j++) {
                if ((ok >> j) & 1u) {
                    const uint32_t d = digit1h(hash_of_stored<K>(keys[j], p), p);
                    const uint32_t pos = atomicAdd(&sm.start[d], 1u);
                    sm.sorted[pos] = keys[j];
                }
            }
            KT_PH(5);
#pragma unroll
            for (int j = 0; j < PB; j++) {
                if (rc[j]) {
                    // what a run lacks to its last group is the empty key (nobody else writes these slots: the placement's
                    // cursor stops at the run's last key)
                    for (uint32_t e = rc[j]; e & (GK - 1u); e++) sm.sorted[rs[j] + e] = EMPTY;
                    // (a stream far past its room - a sender's heavy-hitter bucket - must not wrap back into it)
                    const uint32_t q0 = got[j] > 0xE0000000ull ? 0xE0000000u : (uint32_t)got[j];
                    sm.delta[tl * PB + j] = q0 - rs[j];
                    if (xcd_place<K>((uint64_t)q0 + rc[j] - 1u, xset, p.nxs) >= p.cap1) {
                        sm.ovf = 1;
                        atomicOr(ovf, 1u);
                    }
                }
            }
            ktd::lds_barrier();
            KT_PH(6);
            stop = sm.ovf != 0;  // the same for every thread
            const bool nxt = q == NQ - 1 && g0 + stride < n_units;  // (workgroup uniform)
            if (nxt) {  // (the keys are placed: their registers are free)
                uint32_t tq = tl;  // (opaque again: what the prefetch derives from the thread index is made here, not kept from the round's top)
                asm volatile("" : "+v"(tq));
                src.prefetch_issue(pre, unit_of(g0 + stride), first_next, unit_of(g0 + 2 * stride), tq % BLOCK, dump);
            }
            bool skip = stop;
#if KT_ABLATION
            if (p.dbg & 0x3000u) skip = true;  // (timing only - 0x1000: no copy-out at all; 0x2000: its stores dropped)
#endif
            raw4 *const mydump = reinterpret_cast<raw4 *>(dump) + tl;
            // group m of the round = slots [m GK, m GK + GK) of sorted[]: one bucket's keys (runs start on group boundaries),
            // the first of them a key, what follows a pad at most
            auto group_at = [&](uint32_t m, bool &live, uint32_t &d, uint64_t &at) {
                const uint32_t i = m << GSH;
                live = i < nk;
                const K first = sm.sorted[live ? i : 0u];
                d = digit1h(hash_of_stored<K>(first, p), p);  // (one shift of a stored hash; 32-bit keys are hashed again)
                at = xcd_place<K>((uint64_t)(uint32_t)(sm.delta[d] + i), xset, p.nxs);
            };
            if (!skip) {
                // copy-out, NST groups per thread: one 16-byte LDS read, one 16-byte store each
#pragma unroll
                for (int u = 0; u < NST; u++) {
                    const uint32_t m = tl + (uint32_t)u * T;
                    bool live;
                    uint32_t d;
                    uint64_t at;
                    group_at(m, live, d, at);
                    const raw4 v = *reinterpret_cast<const raw4 *>(&sm.sorted[live ? m << GSH : 0u]);
                    const bool fits = live && at < p.cap1;
                    raw4 *const dst = fits ? reinterpret_cast<raw4 *>(keys1 + ((uint64_t)d * p.cap1 + at)) : mydump;  // (one store on every path)
                    *dst = v;
                }
                if (nxt) src.template prefetch_take<NST>(tk, pre);
            } else {
#if KT_ABLATION
                if ((p.dbg & 0x2000u) && !stop) {  // (the copy-out's LDS side without its stores)
                    uint32_t acc = 0;
#pragma unroll
                    for (int u = 0; u < NST; u++) {
                        bool live;
                        uint32_t d;
                        uint64_t at;
                        group_at(tl + (uint32_t)u * T, live, d, at);
                        acc += d + (uint32_t)at;
                    }
                    if (acc == 0x1234567u) mydump->x = acc;
                }
#endif
                if (nxt) src.template prefetch_take<0>(tk, pre);
            }
            if (nxt) first_cur = first_next;
            // the next round's count, scan and layout touch nothing the copy-out reads (delta[] is written behind the next
            // placement, two barriers on); only the staging of a new segment (it shares LDS with the sort buffer) has to wait
            if (q == NQ - 1) ktd::lds_barrier();
            KT_PH(7);
        }
    }
#if KT_ABLATION
    if (tid == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&kt_dbg_phase[8 + i], (unsigned long long)phs[i]);
#endif
}

// ---- scatter1y: two sort buffers, the copy-out of a round spread over the whole of the next one -----------------------------
// scatter1x's phases add up (section 4.2 of DESIGN.md, round 5): its compute phases issue no stores and its copy-out does
// nothing else, and a CU cannot issue stores faster than ~7 bytes per clock - a round's 128 KB hold the CU for 7 us.  Here
// a round is HALF the size (8 K 64-bit / 16 K 32-bit keys: PER2 window starts per thread) and there are TWO sort buffers:
// the groups of round r leave one 16-byte store at a time from five places inside round r + 1 - the walk, the scan, the
// placement - so the store queue drains under everything else the CU does, and nobody waits for it.  delta[] follows the
// buffers (one per parity: the round before reads its own while this round writes its); the round's counters are 16-bit,
// two to a word - that is what makes room for the second delta[] at 64-bit keys and 1024 buckets; the staged segments
// borrow the buffer that the round about to begin will fill (its former content left the CU a round ago).  LDS: 2 x (T x
// PER2 + B1 x (GK - 1)) keys + 14 KB - carved at run time (the pads follow B1); a shape that does not fit 160 KB keeps
// scatter1x.
#ifndef KT_S1Y_CARRY
#define KT_S1Y_CARRY 1   // 0: a bucket's odd key leaves with an empty key beside it, every round (A/B builds)
#endif
#ifndef KT_S1Y_BATCH
#define KT_S1Y_BATCH 1   // 0: the placement asks for a cursor and waits for it, key by key; 2: batched with 64-bit keys too (A/B builds)
#endif
// (ctr k=15, 16 keys per thread and round: level 1 15.2 -> 12.9 ms; 64-bit keys, 8 per thread: 10.38 -> 10.30 ms over packed reads,
// 2.31 -> 2.37 ms per piece over records - so 32-bit keys only)
template <class K>
constexpr bool s1y_batch() { return KT_S1Y_BATCH == 2 || (KT_S1Y_BATCH == 1 && sizeof(K) == 4); }
template <class K>
constexpr int half_per() { return wide_per<K>() / 2; }
template <class K, int T>
struct Scatter1YShared {
    K *sorted[2];
    // the round's counters: [MAX_B1], or - 64-bit keys, where 2 KB decide whether the second delta[] fits - two 16-bit ones to
    // a word, [MAX_B1 / 2] (a round has 8 K keys; two buckets per word cost the count's atomics more conflicts: at k=15, 256
    // buckets, the packed form measured 19.5 against 19.0 ms, so 32-bit keys keep a word per bucket)
    static constexpr bool PACK = sizeof(K) == 8;
    uint32_t *cnt2;
    uint32_t *start;  // [MAX_B1] the runs of the round in sorted[]: the scan's result, the placement's cursors
    uint32_t *delta[2];  // [MAX_B1] per parity: (the run's first place in its set's stream of the bucket) - (its start in sorted[])
    uint32_t *tmp, *ovf;
    __host__ __device__ static constexpr size_t slots(uint32_t B1) { return ((size_t)T * half_per<K>() + (size_t)B1 * (group_keys<K>() - 1) + 63) / 64 * 64; }
    static constexpr size_t bytes(uint32_t B1) { return 2 * slots(B1) * sizeof(K) + MAX_B1 * (sizeof(K) == 8 ? 2 : 4) + 3 * MAX_B1 * 4 + 16 * 4 + 16; }
    __device__ Scatter1YShared(unsigned char *raw, uint32_t B1) {
        sorted[0] = reinterpret_cast<K *>(raw);
        sorted[1] = sorted[0] + slots(B1);
        cnt2 = reinterpret_cast<uint32_t *>(sorted[1] + slots(B1));
        start = cnt2 + (PACK ? MAX_B1 / 2 : MAX_B1);
        delta[0] = start + MAX_B1;
        delta[1] = delta[0] + MAX_B1;
        tmp = delta[1] + MAX_B1;
        ovf = tmp + 16;
    }
};
static_assert(Scatter1YShared<uint64_t, 1024>::bytes(MAX_B1) <= 160 * 1024, "64-bit keys, 1024 buckets: the largest shape level 1 is asked for");

template <class Source, class K, int T>
__global__ __launch_bounds__(T, 4) void scatter1y_kernel(Source src, Plan p, unsigned long long *__restrict__ xcur,
                                                      uint32_t *__restrict__ ovf, K *__restrict__ keys1,
                                                      unsigned long long *__restrict__ dump) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const Scatter1YShared<K, T> sm(smem_raw, p.B1);
    constexpr int PER = half_per<K>(), NQ = ktseg::PER_THREAD / PER, GROUPS = T / BLOCK, PB = MAX_B1 / T;
    constexpr uint32_t GK = group_keys<K>(), GSH = GK == 2 ? 1 : 2;
    constexpr int NST = (int)(((size_t)T * PER + (size_t)MAX_B1 * (GK - 1)) / GK + T - 1) / T;  // 16-byte stores per thread and round
    static_assert(NST == 5 && PB == 1, "the five places the stores leave from");
    static_assert(GROUPS * sizeof(SegShared) <= (size_t)T * PER * sizeof(K), "the staged segments fit a sort buffer");
    constexpr K EMPTY = empty_of<K>();
    typedef uint32_t raw4 __attribute__((ext_vector_type(4)));
    const uint32_t tid = threadIdx.x, grp = tid / BLOCK, t = tid % BLOCK;
    const uint32_t xset = (p.xmap >> (4u * xcc_id())) & 7u;
    unsigned long long *const mycur = xcur + (size_t)xset * p.B1;
    constexpr bool PACK = Scatter1YShared<K, T>::PACK;
    if (tid < (PACK ? MAX_B1 / 2 : MAX_B1)) sm.cnt2[tid] = 0;  // (every round leaves the counters zeroed for the next one)
    if (tid == 0) *sm.ovf = 0;
    const uint64_t n_units = src.n_units();
    bool stop = false;
    const uint64_t stride = (uint64_t)gridDim.x * GROUPS;
    auto unit_of = [&](uint64_t g0) { return g0 + grp < n_units ? g0 + grp : n_units - 1; };
    typename Source::Pre2 pre;
    typename Source::Taken tk{};
    uint64_t first_cur = 0;
    {
        const uint64_t gfirst = (uint64_t)blockIdx.x * GROUPS;
        if (gfirst < n_units) {
            first_cur = src.first_of(unit_of(gfirst));
            src.prefetch_issue(pre, unit_of(gfirst), first_cur, unit_of(gfirst + stride), t, dump);
            src.template prefetch_take<0>(tk, pre);
        }
    }
    uint32_t par = 0;      // parity of the round being made: it fills sorted[par], the round before it lies in sorted[par ^ 1]
    uint32_t nk_prev = 0;  // slots of the round whose groups are still leaving (0: none)
    // CARRY (64-bit keys: two to a group): a bucket's odd key does not leave with an empty key beside it - 7 % of what level 1
    // wrote and level 2 read at 1024 buckets - but waits in the sort buffer it lies in (which stays as it is for the whole of
    // the next round: it is the one being copied out) and is counted and placed again by the bucket's thread in the next round.
    // The copy-out knows the group by the empty key in its second place and sends it to the dump line; the space taken in the
    // stream is the run's even part; what is left at the end of the launch goes out key by key, an empty key beside it.
    constexpr bool CARRY = KT_S1Y_CARRY && GK == 2;
    uint32_t rs_prev = 0, rc_prev = 0;  // this thread's bucket in the round before: where its run starts in sorted[], its keys
    constexpr uint32_t LSH1 = sizeof(K) == 8 ? 4 : 5;  // keys per 128-byte line (xcd_place)
    const uint32_t cap1_32 = (uint32_t)p.cap1, cap_lines = (uint32_t)(p.cap1 >> LSH1);  // (plan_job: a region is < 2^31 bytes)
    // group u of this thread's share of the round before: out to its place, or - nothing there, or no room - to the dump line
    auto emit = [&](int u, uint32_t tl, uint32_t pp) {
        uint32_t m = tl + (uint32_t)u * T;
        asm volatile("" : "+v"(m));  // (pinned to its place in the round)
        const uint32_t i = m << GSH;
        bool live = i < nk_prev;
        const K *const sb = sm.sorted[pp];
        const raw4 v = *reinterpret_cast<const raw4 *>(&sb[live ? i : 0u]);
        if constexpr (CARRY) live = live && (v[2] & v[3]) != 0xFFFFFFFFu;  // (a run's odd key: it stays)
        const K first = sb[live ? i : 0u];
        const uint32_t d = digit1h(hash_of_stored<K>(first, p), p);
        // xcd_place in 32 bits (the line index of a stream that ran 2^32 keys past its room still fits), one 64-bit multiply-add
        // for the place in the key array: the general form - two 64-bit shifts, a 64-bit compare, two multiply-adds - is a
        // fifth of the kernel's issue cycles, five times per thread and round
        const uint32_t pos = sm.delta[pp][d] + i;
        const uint32_t lx = ((pos >> LSH1) << p.nxs) | xset;
        const bool room = lx < cap_lines;
        const uint32_t at = (lx << LSH1) | (pos & ((1u << LSH1) - 1u));
        const bool fits = live && room;
        raw4 *const dst = fits ? reinterpret_cast<raw4 *>(keys1 + ktd::mad64(d, cap1_32, at)) : reinterpret_cast<raw4 *>(dump) + tl;
        *dst = v;
    };
#if KT_ABLATION
    unsigned long long tph = __builtin_readcyclecounter();
    uint32_t phs[8] = {};
#endif
    for (uint64_t g0 = (uint64_t)blockIdx.x * GROUPS; g0 < n_units && !stop; g0 += stride) {
        const bool valid = g0 + grp < n_units;  // (a group past the end walks the last unit and keeps nothing)
        // the staged segments borrow the buffer the round about to begin will fill
        SegShared *const segs = reinterpret_cast<SegShared *>(sm.sorted[par]);
        auto wk = src.open_taken(unit_of(g0), first_cur, segs[grp], t, tk);
        const uint64_t first_next = tk.first_next;
        KT_PH(0);
#pragma unroll
        for (int q = 0; q < NQ; q++) {  // (no early exit in here: the loop must stay unrolled)
            uint32_t tl = tid;  // (opaque per round: see scatter1x_kernel)
            asm volatile("" : "+v"(tl));
            const uint32_t pp = par ^ 1u;
            const bool nxt = q == NQ - 1 && g0 + stride < n_units;  // (workgroup uniform)
            // the next unit's reads leave at the top of the unit's last round and are taken at its end, behind the five stores
            // and the cursor atomic that follow them on every path: s_waitcnt vmcnt(NST + 1)
            if (nxt) {
                uint32_t tq = tl;
                asm volatile("" : "+v"(tq));
                src.prefetch_issue(pre, unit_of(g0 + stride), first_next, unit_of(g0 + 2 * stride), tq % BLOCK, dump);
            }
            bool skip = stop;
#if KT_ABLATION
            if (p.dbg & 0x1000u) skip = true;  // (timing only: no copy-out at all)
#endif
            // where the five stores of the round before leave from: the middle and the end of the count, behind the scan, the
            // middle and the end of the placement - every wave at the same steps, three of them just in front of a barrier
            // (measured, k=31 / k=15: 13.7 / 19.0 ms; none of them next to a barrier 14.3-14.6 / 18.8; every wave's stores at
            // steps of its own, spread against the other waves' 15.0 / 18.8 - a wave held up at the store queue in front of
            // a barrier holds nobody up who is not waiting there anyway)
            auto emit_at = [&](int slot) {
                const int u = slot == PER / 2 - 1 ? 0 : slot == PER - 1 ? 1 : slot == PER + PER / 2 ? 3 : slot == 2 * PER - 1 ? 4 : -1;
                if (!skip && u >= 0) emit(u, tl, pp);
            };
            K keys[PER];
            uint32_t ok;
            src.template take<PER, K>(wk, tl % BLOCK, keys, ok);
            if (!valid) ok = 0;
#pragma unroll
            for (int j = 0; j < PER; j++) {  // (the keys take their stored form here, where the first digit is needed)
                keys[j] = to_stored<K>((uint64_t)keys[j]);
                if constexpr (s1y_batch<K>()) {  // (a key that does not count adds 0: no branch, no exec-mask bookkeeping per key)
                    const uint32_t d = digit1h(hash_of_stored<K>(keys[j], p), p), one = (ok >> j) & 1u;
                    if constexpr (PACK) atomicAdd(&sm.cnt2[d >> 1], one << ((d & 1u) * 16u));
                    else atomicAdd(&sm.cnt2[d], one);
                } else if ((ok >> j) & 1u) {
                    const uint32_t d = digit1h(hash_of_stored<K>(keys[j], p), p);
                    if constexpr (PACK) atomicAdd(&sm.cnt2[d >> 1], 1u << ((d & 1u) * 16u));
                    else atomicAdd(&sm.cnt2[d], 1u);
                }
                emit_at(j);
            }
            K carried = EMPTY;
            uint32_t has_carry = 0;
            if constexpr (CARRY) {
                has_carry = rc_prev & 1u;
                carried = sm.sorted[pp][has_carry ? rs_prev + rc_prev - 1u : 0u];
                if constexpr (PACK) atomicAdd(&sm.cnt2[tl >> 1], has_carry << ((tl & 1u) * 16u));  // (has_carry: tl is a bucket)
                else atomicAdd(&sm.cnt2[tl], has_carry);
            }
            KT_PH(1);
            ktd::lds_barrier();
            KT_PH(2);
            uint32_t *const xs = sm.start;
            auto count_of = [&](uint32_t d) { return PACK ? (sm.cnt2[d >> 1] >> ((d & 1u) * 16u)) & 0xFFFFu : sm.cnt2[d]; };
            // (no closing barrier: the thread reads back its own start only - PB = 1, tl is the thread's index - and the
            // barrier in front of the placement stands between the scan and everybody else's)
            const uint32_t nk = block_excl_scan_f<T, PB, GK - 1, false>(count_of, xs, p.B1, sm.tmp);
            // (no branch around the atomic: see scatter1x_kernel)
            const uint32_t d0 = tl, dc = d0 & (p.B1 - 1u);
            const uint32_t rc = d0 < p.B1 ? count_of(dc) : 0u;
            const uint32_t rs = xs[dc];
            // for the next round's count (an atomic: the word's other half is its neighbour's)
            if (d0 < p.B1) {
                if constexpr (PACK) atomicAnd(&sm.cnt2[d0 >> 1], (d0 & 1u) ? 0x0000FFFFu : 0xFFFF0000u);
                else sm.cnt2[d0] = 0;
            }
            const uint32_t take = CARRY ? rc & ~1u : (rc + GK - 1u) & ~(GK - 1u);  // the keys of the run that leave
            const unsigned long long got = __hip_atomic_fetch_add(&mycur[dc], (unsigned long long)take,
                                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!skip) emit(2, tl, pp);
            KT_PH(3);
            ktd::lds_barrier();  // start[] was read above; the placement pass uses it as its cursors
            KT_PH(4);
            K *const sb = sm.sorted[par];
            if constexpr (s1y_batch<K>()) {
                // the placement's cursor atomics leave together, all PER of them, and the keys follow as the answers come in: a
                // key that does not count adds 0 to its cursor (no branch around the atomic: under one the compiler waits for
                // every answer where it is asked for - PER LDS round trips one after the other, four waves per SIMD to hide them)
                uint32_t at[PER];
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    const uint32_t d = digit1h(hash_of_stored<K>(keys[j], p), p);
                    at[j] = atomicAdd(&xs[d], (ok >> j) & 1u);
                }
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    if ((ok >> j) & 1u) sb[at[j]] = keys[j];
                    emit_at(PER + j);
                }
            } else {
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    if ((ok >> j) & 1u) {
                        const uint32_t d = digit1h(hash_of_stored<K>(keys[j], p), p);
                        const uint32_t pos = atomicAdd(&xs[d], 1u);
                        sb[pos] = keys[j];
                    }
                    emit_at(PER + j);
                }
            }
            if constexpr (CARRY) {  // the bucket's odd key of the round before: placed like any other key of its bucket
                const uint32_t posc = atomicAdd(&xs[dc], has_carry);
                if (has_carry) sb[posc] = carried;
            }
            KT_PH(5);
            if (rc) {
                for (uint32_t e = rc; e & (GK - 1u); e++) sb[rs + e] = EMPTY;  // what the run lacks to its last group
                // (a stream far past its room - a sender's heavy-hitter bucket - must not wrap back into it)
                const uint32_t q0 = got > 0xE0000000ull ? 0xE0000000u : (uint32_t)got;
                sm.delta[par][tl] = q0 - rs;
                const uint32_t leaving = CARRY ? take : rc;
                if (leaving && xcd_place<K>((uint64_t)q0 + leaving - 1u, xset, p.nxs) >= p.cap1) {
                    *sm.ovf = 1;
                    atomicOr(ovf, 1u);
                }
            }
            rs_prev = rs;
            rc_prev = rc;
            if (nxt) {
                if (!skip) src.template prefetch_take<NST + 1>(tk, pre);
                else src.template prefetch_take<0>(tk, pre);
                first_cur = first_next;
            }
            ktd::lds_barrier();
            KT_PH(6);
            stop = *sm.ovf != 0;  // the same for every thread
            nk_prev = nk;
            par ^= 1u;
            KT_PH(7);
        }
    }
    // the last round's groups
    if (!stop && nk_prev) {
        uint32_t tl = tid;
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int u = 0; u < NST; u++) emit(u, tl, par ^ 1u);
    }
    if constexpr (CARRY) {
        // what the last round left waiting: one group per bucket with an odd key, an empty key beside it
        if (!stop && (rc_prev & 1u)) {
            const K key = sm.sorted[par ^ 1u][rs_prev + rc_prev - 1u];
            const uint32_t d = tid & (p.B1 - 1u);
            const unsigned long long got = __hip_atomic_fetch_add(&mycur[d], 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t pos = got > 0xE0000000ull ? 0xE0000000u : (uint32_t)got;
            const uint32_t lx = ((pos >> LSH1) << p.nxs) | xset;
            if (lx < cap_lines) {
                const uint32_t at = (lx << LSH1) | (pos & ((1u << LSH1) - 1u));
                K *const dst = keys1 + ktd::mad64(d, cap1_32, at);
                dst[0] = key;
                dst[1] = EMPTY;
            } else {
                atomicOr(ovf, 1u);
            }
        }
    }
#if KT_ABLATION
    if (tid == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&kt_dbg_phase[8 + i], (unsigned long long)phs[i]);
#endif
}

// after the last source of a job: a bucket's region is read up to its EXTENT - the lines of the fullest
// cursor set, times the sets - so what the other sets have not filled of their lines up to there gets the empty key
// (level 2 skips it), and the extent goes where level 2 looks for the bucket's key count
template <class K>
__global__ __launch_bounds__(BLOCK) void xcd_tails_kernel(Plan p, const unsigned long long *__restrict__ xcur,
                                                          uint64_t *__restrict__ extent, K *__restrict__ keys1) {
    constexpr uint32_t LSH = sizeof(K) == 8 ? 4 : 5, LK = 1u << LSH;
    const uint32_t NX = 1u << p.nxs;
    const uint64_t set_room = (p.cap1 >> (LSH + p.nxs)) << LSH;  // keys a set can place in a region
    for (uint32_t d = blockIdx.x; d < p.B1; d += gridDim.x) {
        uint64_t c[8], top = 0;
        for (uint32_t x = 0; x < NX; x++) {
            const uint64_t v = xcur[(size_t)x * p.B1 + d];
            c[x] = v < set_room ? v : set_room;  // (what went past the room was parked or made the pass stop)
            const uint64_t lines = (c[x] + LK - 1) >> LSH;
            top = lines > top ? lines : top;
        }
        for (uint32_t x = 0; x < NX; x++)
            for (uint64_t q = c[x] + threadIdx.x; q < (top << LSH); q += BLOCK)
                keys1[(uint64_t)d * p.cap1 + xcd_place<K>(q, x, p.nxs)] = empty_of<K>();
        if (threadIdx.x == 0) extent[d] = (top << LSH) << p.nxs;
    }
}

// ---- part2: one workgroup per level-1 bucket -----------------------------------------------------------
// LDS of part2, carved at run time so that the per-digit arrays take B2 entries, not MAX_B: with 32-bit keys and
// B2 <= 1024 two workgroups fit a CU
template <class K, bool BIG>
struct Part2Shared {
    K *sorted;         // [chunk2<K, BIG>() + 1] (+ a spare slot where the placement drops empty keys)
    uint32_t *cur;     // [B2] where the fine bucket's next key goes, relative to the level-1 bucket's first key
    uint32_t *cnt;     // [2][B2] keys of the chunk per fine bucket.  Two of them, used by alternate chunks: the next chunk
                       // counts while this one is copied out.
    uint32_t *start;   // [B2] the chunk's runs in sorted[]
    uint32_t *delta;   // [B2] cur - start: what the copy-out adds to a sorted key's index to get its place in the bucket
    uint32_t *tmp;     // [16] block scan scratch
    uint32_t *flag;    // [1] (+1 pad) fixed fine regions: a fine bucket is outgrowing its room
    uint16_t *sdig;    // [chunk2<K, BIG>() + 1]
    // 20 bytes per fine bucket
    static size_t bytes(uint32_t B2) {
        return (size_t)chunk2<K, BIG>() * sizeof(K) + 16 + (size_t)B2 * 20 + 16 * 4 + 8 +
               (p2_sdig<K, BIG>() ? (size_t)chunk2<K, BIG>() * 2 + 16 : 0);
    }
    __device__ Part2Shared(unsigned char *raw, uint32_t B2) {
        sorted = reinterpret_cast<K *>(raw);
        raw += (size_t)chunk2<K, BIG>() * sizeof(K) + 16;
        cur = reinterpret_cast<uint32_t *>(raw);
        cnt = cur + B2;
        start = cnt + 2 * B2;
        delta = start + B2;
        tmp = delta + B2;
        flag = tmp + 16;
        sdig = reinterpret_cast<uint16_t *>(flag + 2);
    }
};

// FIXED (with paged level 1): every fine bucket owns cap2 keys of room in keys2, so the whole-bucket histogram
// pass (a second read of keys1) is not needed - as long as the bucket's keys spread over its fine buckets the way
// a hash spreads distinct keys.  They do not when few distinct k-mers make up the batch (deep coverage of a small
// genome, repeats): the attempt watches every fine bucket's fill against the fraction of the level-1 bucket
// processed so far, and as soon as one is heading past its room it stops placing keys, finishes the pass counting
// only (which is exactly the histogram), and the bucket is redone with the exact boundaries - the price of a
// wrong guess is the part of the pass done before it was noticed.
// Where part2 reads a bucket from: its region of the level-1 output - bucket jl = srcs[0].keys + jl * cap1,
// min(srcs[0].counts[jl], cap1) keys, empty keys in the gaps (the kernels walk a list of such sources; a job has one).
// (exact level 1, srcs == nullptr: keys1 is dense, bucket j = [bstart[j], bstart[j + 1]).)
#ifndef KT_P2_LOAD16
#define KT_P2_LOAD16 0  // two keys per 16-byte load in part2: measured 16.0 against 15.5 ms (k=31) - off
#endif
struct P2In {
    const void *keys1;
    const uint64_t *bstart;
    const kt_seg_src *srcs;
    uint32_t n_src;
    __device__ __forceinline__ uint64_t out_of(uint32_t jl, uint64_t room1) const { return (uint64_t)jl * room1; }
};

template <class K, bool FIXED, bool BIG, bool L16 = false>
__global__ __launch_bounds__((p2t<K, BIG>()), (p2t<K, BIG>() == 1024 ? 4 : 2)) void part2_kernel(P2In in, Plan p,
                                                      K *__restrict__ keys2, uint64_t *__restrict__ fstart,
                                                      uint64_t *__restrict__ fend, const uint32_t *__restrict__ only) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const Part2Shared<K, BIG> sm(smem_raw, p.B2);
    constexpr K EMPTY = empty_of<K>();
    constexpr int P2T = p2t<K, BIG>();
    constexpr int PER = chunk2<K, BIG>() / P2T;  // 16 (32) keys per thread, held in registers
    constexpr uint32_t CH = chunk2<K, BIG>();
    const uint32_t tid = threadIdx.x;
    const uint32_t nd = p.B1;
    for (uint32_t jl = blockIdx.x; jl < nd; jl += gridDim.x) {
        if (only && !only[jl]) continue;  // (the redo pass behind part2_fast_kernel: only the buckets whose attempt failed)
        // the bucket's segments, and where the bucket lives in keys2
        const uint32_t n_seg = in.srcs ? in.n_src : 1u;
        auto segment = [&](uint32_t sidx, const K *&base, uint64_t &n) {
            if (in.srcs) {
                const kt_seg_src &q = in.srcs[sidx];
                const uint64_t c = q.counts[jl];
                base = reinterpret_cast<const K *>(q.keys) + (uint64_t)jl * q.cap1;
                n = c < q.cap1 ? c : q.cap1;
            } else {
                base = reinterpret_cast<const K *>(in.keys1) + in.bstart[jl];
                n = in.bstart[jl + 1] - in.bstart[jl];
            }
        };
        uint64_t total = 0;
        for (uint32_t sidx = 0; sidx < n_seg; sidx++) {
            const K *b;
            uint64_t n;
            segment(sidx, b, n);
            total += n;
        }
        const uint64_t lo = in.srcs ? in.out_of(jl, p.room1) : in.bstart[jl];
        auto load_chunk = [&](const K *base, uint64_t n, uint64_t c0, K (&dst)[PER]) {
            if constexpr (sizeof(K) == 8 && L16) {  // two keys per load (regions are 16-byte aligned: paged level 1)
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int u = 0; u < PER / 2; u++) {
                    const uint64_t i = c0 + ((uint64_t)u * P2T + tid) * 2;
                    u64x2 v = {(unsigned long long)EMPTY, (unsigned long long)EMPTY};
                    if (i + 1 < n) v = *reinterpret_cast<const u64x2 *>(base + i);
                    else if (i < n) v.x = base[i];
                    dst[2 * u] = (K)v.x;
                    dst[2 * u + 1] = (K)v.y;
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const uint64_t i = c0 + (uint64_t)u * P2T + tid;
                dst[u] = i < n ? base[i] : EMPTY;
            }
        };
        // one pass over the bucket in chunks of chunk2<K, BIG>() keys, segment after segment: counting sort in LDS, runs
        // appended at the fine buckets' cursors.  The next chunk's keys are loaded while the current one is sorted.
        // attempt = fixed fine regions (cursors may run past their room: then nothing is stored any more)
        auto run_pass = [&](const bool attempt) {
            // One set of key registers: the next chunk's loads are issued once the current chunk's keys have been placed in
            // LDS (they are dead then) and travel during the copy-out.  (Holding the prefetched chunk beside the current one
            // - 64 registers of keys - spilled a quarter of them at the 128 registers a 16-wave workgroup gets.)
            K kcur[PER];
            bool counting_only = false;
            uint64_t seen_keys = 0;
            // Five barriers per chunk (it was seven): the chunks alternate between two count arrays, each zeroed a chunk
            // ahead, so a wave that has copied its share of the sorted chunk out goes straight on to the cursor update
            // and the next chunk's count while the others are still copying.
            uint32_t par = 0;
            for (uint32_t i = tid; i < 2 * p.B2; i += P2T) sm.cnt[i] = 0;
            ktd::lds_barrier();
            for (uint32_t sidx = 0; sidx < n_seg; sidx++) {
                const K *base;
                uint64_t n;
                segment(sidx, base, n);
                if (n) load_chunk(base, n, 0, kcur);
                for (uint64_t c0 = 0; c0 < n; c0 += CH, par ^= 1u) {
                    uint32_t *const cntc = sm.cnt + par * p.B2, *const cntn = sm.cnt + (par ^ 1u) * p.B2;
#pragma unroll
                    for (int u = 0; u < PER; u++) {
                        const uint32_t d = digit2h(hash_of_stored<K>(kcur[u], p), p);
                        if (kcur[u] != EMPTY) atomicAdd(&cntc[d], 1u);
                    }
                    ktd::lds_barrier();
                    if (attempt) counting_only = *sm.flag != 0;  // (raised in a cursor update before the barrier above)
                    if (!counting_only) {
                        const uint32_t nc = block_excl_scan<P2T>(cntc, sm.start, p.B2, sm.tmp);  // keys in the chunk
                        // (cntc has been summed: from here to the cursor update it holds cur - start, what the copy-out
                        // adds to a sorted key's index to get its place in the level-1 bucket; the other array - last
                        // read by the previous chunk's cursor update, before the barrier above - is zeroed for the next)
                        for (uint32_t i = tid; i < p.B2; i += P2T) {
                            cntc[i] = sm.cur[i] - sm.start[i];
                            cntn[i] = 0;
                        }
                        ktd::lds_barrier();
#pragma unroll
                        for (int u = 0; u < PER; u++) {
                            if (kcur[u] != EMPTY) {
                                const uint32_t d = digit2h(hash_of_stored<K>(kcur[u], p), p);
                                const uint32_t pos = atomicAdd(&sm.start[d], 1u);
                                sm.sorted[pos] = kcur[u];
                                if constexpr (p2_sdig<K, BIG>()) sm.sdig[pos] = (uint16_t)d;
                            }
                        }
                        if (c0 + CH < n) load_chunk(base, n, c0 + CH, kcur);  // the next chunk travels during the copy-out
                        ktd::lds_barrier();
                        for (uint32_t i = tid; i < nc; i += P2T) {
                            const K key = sm.sorted[i];
                            uint32_t d;
                            if constexpr (p2_sdig<K, BIG>()) d = sm.sdig[i];
                            else d = digit2h(hash_of_stored<K>(key, p), p);
                            const uint64_t pos = lo + (uint32_t)(cntc[d] + i);
                            if (!attempt || pos < lo + (uint64_t)(d + 1) * p.cap2) keys2[pos] = key;
                        }
                    } else {
                        if (c0 + CH < n) load_chunk(base, n, c0 + CH, kcur);
                        for (uint32_t i = tid; i < p.B2; i += P2T) cntn[i] = 0;
                    }
                    // cursors move on (no barrier: nothing the copy-out of the other waves reads is written here); during
                    // an attempt every fine bucket's fill is held against its room scaled to the part of the level-1
                    // bucket seen so far (+ 6 sigma): hashed distinct keys never get there
                    float allowed = 0.f;
                    if (attempt) {
                        const float seen = (float)(seen_keys + c0 + CH) / (float)total;
                        // (the bucket may hold fewer keys than its room: what counts is the share of a fine bucket)
                        const float room = (float)p.cap2 * (seen < 1.f ? seen : 1.f);
                        allowed = 0.93f * room + 6.f * sqrtf(room) + 32.f;
                    }
                    for (uint32_t i = tid; i < p.B2; i += P2T) {
                        // placed: start[] stands at the runs' ends, so start + (cur - start at their beginnings) = cur + count
                        const uint32_t c = counting_only ? sm.cur[i] + cntc[i] : sm.start[i] + cntc[i];
                        sm.cur[i] = c;
                        if (attempt && !counting_only && (float)(c - i * (uint32_t)p.cap2) > allowed) *sm.flag = 1;
                    }
                    if (counting_only) ktd::lds_barrier();  // (the zeroing above against the next chunk's count)
                }
                seen_keys += n;
            }
            ktd::lds_barrier();
        };

        bool exact = !FIXED;
        if constexpr (FIXED) {
            for (uint32_t i = tid; i < p.B2; i += P2T) sm.cur[i] = i * (uint32_t)p.cap2;  // (room1 = B2 * cap2 < 2^32)
            if (tid == 0) *sm.flag = 0;
            ktd::lds_barrier();
            run_pass(true);
            // did every fine bucket stay inside its room?  (the running check is a prediction; this is the fact)
            for (uint32_t i = tid; i < p.B2; i += P2T)
                if (sm.cur[i] - i * (uint32_t)p.cap2 > p.cap2) *sm.flag = 1;
            ktd::lds_barrier();
            if (*sm.flag == 0) {
                for (uint32_t i = tid; i < p.B2; i += P2T) {
                    fstart[(uint64_t)jl * p.B2 + i] = lo + (uint64_t)i * p.cap2;
                    fend[(uint64_t)jl * p.B2 + i] = lo + sm.cur[i];
                }
            } else {  // no: the cursors hold the exact sizes now
                for (uint32_t i = tid; i < p.B2; i += P2T) sm.cnt[i] = sm.cur[i] - i * (uint32_t)p.cap2;
                exact = true;
            }
            ktd::lds_barrier();
        } else {
            // whole-bucket histogram of d2
            for (uint32_t i = tid; i < p.B2; i += P2T) sm.cnt[i] = 0;
            ktd::lds_barrier();
            for (uint32_t sidx = 0; sidx < n_seg; sidx++) {
                const K *base;
                uint64_t n;
                segment(sidx, base, n);
                for (uint64_t i0 = tid; i0 < n; i0 += (uint64_t)P2T * 8) {  // 8 loads in flight per thread
                    K kk[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint64_t i = i0 + (uint64_t)u * P2T;
                        kk[u] = i < n ? base[i] : EMPTY;
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (kk[u] != EMPTY) atomicAdd(&sm.cnt[digit2h(hash_of_stored<K>(kk[u], p), p)], 1u);
                }
            }
            ktd::lds_barrier();
        }
        if (exact) {
            // fine bucket boundaries from the sizes (a level-1 bucket of > 4 G keys is not supported)
            block_excl_scan<P2T>(sm.cnt, sm.start, p.B2, sm.tmp);
            for (uint32_t i = tid; i < p.B2; i += P2T) {
                const uint64_t pos = lo + sm.start[i];
                sm.cur[i] = sm.start[i];
                fstart[(uint64_t)jl * p.B2 + i] = pos;
                fend[(uint64_t)jl * p.B2 + i] = pos + sm.cnt[i];
            }
            ktd::lds_barrier();
            run_pass(false);
        }
    }
}

// ---- part2, the kernel the bulk path runs (round 4): fixed fine regions, nothing else --------------------------------
// part2_kernel<.., FIXED> with everything that is not the common case taken out: one attempt with fixed fine regions
// per bucket; a bucket whose keys do not spread like hashed distinct keys raises fail[jl] and is left to a second launch of
// the general kernel (exact boundaries from a histogram pass), which skips every other bucket.  What round 3's ISA and
// ablations showed about the old loop, and what this one does about it:
//  * the wave waited for every LDS round trip on its own: sixteen times "ds_add_rtn, s_waitcnt lgkmcnt(0), ds_write_b64"
//    in the placement, a rolled "ds_read_b64, wait, ds_read_b32, wait, store" in the copy-out - with 4 waves per SIMD
//    nothing hid those ~50 round trips per chunk.  Now ONE LDS atomic per key: the count's atomic returns the key's rank
//    inside its run, so its place is start[d] + rank - a plain read, identical for the keys of a run - and every phase
//    issues its LDS operations for a batch of keys before it waits once.
//  * the stores were the other half (13.8 ms -> 6.7 ms with the stores ablated): they sat inside exec-mask branches behind
//    the next chunk's loads, so the wait for those loads at the top of the next chunk had to be vmcnt(0) - every wave
//    drained its stores once per chunk, and the one workgroup of a CU does nothing else meanwhile.  Now loads and stores
//    are issued out of the compiler's sight (buf_load8_async / buf_store_async) and the wait is vmcnt(PER): the loads are in,
//    this chunk's stores drain while the next chunk is counted, scanned and placed.
//  * registers: 512 threads x 32 keys (64-bit keys) in a 256-register budget, the general kernel's other paths gone, the
//    scan with a compile-time number of counters per thread (PB), the thread index made opaque per chunk - the hot loop
//    must not touch scratch: a reload is a vmcnt(0).
template <class K, bool BIG, int PB>
__global__ __launch_bounds__((p2t<K, BIG>()), (p2t<K, BIG>() == 1024 ? 4 : 2)) void part2_fast_kernel(
    P2In in, Plan p, K *__restrict__ keys2, uint64_t *__restrict__ fstart, uint64_t *__restrict__ fend,
    uint32_t *__restrict__ fail) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const Part2Shared<K, BIG> sm(smem_raw, p.B2);
    constexpr K EMPTY = empty_of<K>();
    constexpr int P2T = p2t<K, BIG>();
    constexpr int PER = chunk2<K, BIG>() / P2T;
    constexpr uint32_t CH = chunk2<K, BIG>();
    constexpr bool RKD = p2_sdig<K, BIG>();  // the digit is kept beside the rank (32-bit keys: it would cost a second hash)
    const uint32_t tid = threadIdx.x;
    const uint32_t nd = p.B1;
    const uint32_t B2 = p.B2, cap2 = (uint32_t)p.cap2, dshift = 32 - p.b1 - p.b2;  // (in the hash's high word: digit2h)
    auto digit = [&](K stored) -> uint32_t { return ((uint32_t)(hash_of_stored<K>(stored, p) >> 32) >> dshift) & (B2 - 1u); };
    for (uint32_t jl = blockIdx.x; jl < nd; jl += gridDim.x) {
        const uint32_t n_seg = in.n_src;
        auto segment = [&](uint32_t sidx, const K *&base, uint64_t &n) {
            const kt_seg_src &q = in.srcs[sidx];
            const uint64_t c = q.counts[jl];
            base = reinterpret_cast<const K *>(q.keys) + (uint64_t)jl * q.cap1;
            n = c < q.cap1 ? c : q.cap1;
        };
        uint64_t total = 0;
        for (uint32_t sidx = 0; sidx < n_seg; sidx++) {
            const K *b;
            uint64_t n;
            segment(sidx, b, n);
            total += n;
        }
        if (fail[jl] != 0) continue;  // (marked already: the exact pass takes it)
        const uint64_t lo = in.out_of(jl, p.room1);
        for (uint32_t i = tid; i < B2; i += P2T) {
            sm.cur[i] = i * cap2;  // (room1 = B2 * cap2 < 2^32)
            sm.cnt[i] = 0;
            sm.cnt[B2 + i] = 0;
        }
        if (tid == 0) *sm.flag = 0;
        ktd::lds_barrier();
        const kt_i32x4 outrs = buf_rsrc(keys2 + lo, (uint32_t)(p.room1 * sizeof(K)));
        // the running check of an attempt: a fine bucket's fill against its room scaled to the keys seen so far, 7 % off,
        // + 6 sigma of a full fine bucket + 32 (bit patterns in scalar registers: a float living across the loop in a
        // vector register is one more candidate for a spill)
        const uint32_t allow_a = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)__float_as_uint(0.93f * (float)cap2 / (float)(total ? total : 1)));
        const uint32_t allow_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(6.f * sqrtf((float)cap2) + 32.f));
        uint64_t seen = 0;
        uint32_t par = 0;
        bool failed = false;
        K kcur[PER];
        uint32_t rk[RKD ? PER : PER / 2];  // ranks: two to a register, or rank | digit << 16
        for (uint32_t sidx = 0; sidx < n_seg && !failed; sidx++) {
            const K *base;
            uint64_t n;
            segment(sidx, base, n);
            if (n == 0) continue;
            auto load_chunk = [&](uint64_t c0) {  // thread t takes keys c0 + u * P2T + t; in flight until buf_wait
                const uint32_t left = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n - c0 > CH ? CH : (uint32_t)(n - c0)));
                const kt_i32x4 rs = buf_rsrc(base + c0, left * (uint32_t)sizeof(K));
                uint32_t voff = tid * (uint32_t)sizeof(K);
#pragma unroll
                for (int u = 0; u < PER; u += 8) buf_load8_async<K, P2T * (int)sizeof(K)>(&kcur[u], rs, voff);
            };
            load_chunk(0);
            buf_wait<0>(kcur);
            for (uint64_t c0 = 0; c0 < n; c0 += CH, par ^= 1u) {
                uint32_t *const cntc = sm.cnt + par * B2, *const cntn = sm.cnt + (par ^ 1u) * B2;
                uint32_t tl = tid;  // (opaque per chunk: nothing derived from the thread index is carried across the loop)
                asm volatile("" : "+v"(tl));
                const uint32_t left = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n - c0 > CH ? CH : (uint32_t)(n - c0)));
                if (left != CH) {  // the segment's last chunk: what lies past its end was read as 0
#pragma unroll
                    for (int u = 0; u < PER; u++) kcur[u] = (uint32_t)u * P2T + tl < left ? kcur[u] : EMPTY;
                }
                // count: the atomic's answer is the key's rank in its fine bucket's run (an empty key - a page gap - adds 0)
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t d = digit(kcur[u]);
                    const uint32_t r = atomicAdd(&cntc[d], kcur[u] != EMPTY ? 1u : 0u);
                    if constexpr (RKD) rk[u] = r | (d << 16);
                    else if (u & 1) rk[u / 2] |= r << 16;
                    else rk[u / 2] = r;
                }
                ktd::lds_barrier();
                if (*sm.flag != 0) {  // (raised by a cursor update before the barrier above: the same for every thread)
                    failed = true;
                    break;
                }
                const uint32_t nc = (uint32_t)__builtin_amdgcn_readfirstlane(
                    (int)block_excl_scan_n<P2T, PB>(cntc, sm.start, B2, sm.tmp));  // keys in the chunk; start[] = the runs
                // (delta is read by the copy-out, behind the barrier after the placement; the other count array - last read
                // by the previous chunk's cursor update - is zeroed for the next chunk)
#pragma unroll
                for (int j = 0; j < PB; j++) {
                    const uint32_t i = tl * PB + j;
                    if (i < B2) {
                        sm.delta[i] = sm.cur[i] - sm.start[i];
                        cntn[i] = 0;
                    }
                }
                constexpr int PLB = 8;  // keys placed at a time: their start[] reads together, then their writes
#pragma unroll
                for (int u0 = 0; u0 < PER; u0 += PLB) {
                    uint32_t at[PLB];
#pragma unroll
                    for (int b = 0; b < PLB; b++) at[b] = sm.start[RKD ? rk[u0 + b] >> 16 : digit(kcur[u0 + b])];
#pragma unroll
                    for (int b = 0; b < PLB; b++) {  // (an empty key goes to the spare slot behind the chunk: no branches)
                        const int u = u0 + b;
                        const uint32_t r = RKD ? rk[u] & 0xFFFFu : (u & 1 ? rk[u / 2] >> 16 : rk[u / 2] & 0xFFFFu);
                        const uint32_t pos = kcur[u] != EMPTY ? at[b] + r : CH;
                        sm.sorted[pos] = kcur[u];
                        if constexpr (RKD) sm.sdig[pos] = (uint16_t)(rk[u] >> 16);
                    }
                }
                bool more = c0 + CH < n;
#if KT_ABLATION
                if (p.dbg & 0x400u) {  // (no loads: synthetic keys)
#pragma unroll
                    for (int u = 0; u < PER; u++) kcur[u] = (K)ktd::mix64(c0 + (uint64_t)u * P2T + tl + ((uint64_t)jl << 40));
                    more = false;
                }
#endif
                if (more) load_chunk(c0 + CH);  // (kcur is dead: placed) the next chunk travels during the copy-out
                ktd::lds_barrier();
                // copy-out, COB keys at a time: LDS reads together, then the stores.  A key past the end of the chunk, or
                // one that finds its fine bucket's room full (the bucket is failing), gets an offset the hardware drops.
                constexpr int COB = 8;
#pragma unroll
                for (int u0 = 0; u0 < PER; u0 += COB) {
                    K kk[COB];
                    uint32_t dd[COB], dl[COB];
                    uint32_t i0 = tl + (uint32_t)u0 * P2T;
                    asm volatile("" : "+v"(i0));
#pragma unroll
                    for (int b = 0; b < COB; b++) {  // (beyond nc: stale keys of an earlier chunk, never stored)
                        kk[b] = sm.sorted[i0 + (uint32_t)b * P2T];
                        if constexpr (RKD) dd[b] = sm.sdig[i0 + (uint32_t)b * P2T];
                    }
#pragma unroll
                    for (int b = 0; b < COB; b++) {
                        if constexpr (!RKD) dd[b] = digit(kk[b]);
                        dl[b] = sm.delta[dd[b] & (B2 - 1u)];
                    }
#pragma unroll
                    for (int b = 0; b < COB; b++) {
                        const uint32_t i = i0 + (uint32_t)b * P2T;
                        const uint32_t rel = dl[b] + i;
                        bool yes = i < nc && rel < (dd[b] + 1u) * cap2;
#if KT_ABLATION
                        if (p.dbg & 0x100u) yes = false;  // (every store dropped by the hardware)
                        if (p.dbg & 0x200u) continue;     // (no store instructions at all)
#endif
                        buf_store_async<K>(kk[b], outrs, yes ? rel * (uint32_t)sizeof(K) : BUF_DROP);
                    }
                }
                // cursors move on (no barrier: nothing the copy-out of the other waves reads is written here)
                {
                    const uint64_t sn = seen + c0 + CH;
                    const float allowed = (float)(uint32_t)(sn < total ? sn : total) * __uint_as_float(allow_a) + __uint_as_float(allow_b);
#pragma unroll
                    for (int j = 0; j < PB; j++) {
                        const uint32_t i = tl * PB + j;
                        if (i < B2) {
                            const uint32_t c = sm.cur[i] + cntc[i];
                            sm.cur[i] = c;
                            if ((float)(c - i * cap2) > allowed) *sm.flag = 1;
                        }
                    }
                }
                // the next chunk's keys are in; this chunk's PER stores stay in flight through the next count / scan / place
                if (more) buf_wait<PER>(kcur);
            }
            seen += n;
        }
        ktd::lds_barrier();
        // did every fine bucket stay inside its room?  (the running check is a prediction; this is the fact)
        if (!failed) {
            for (uint32_t i = tid; i < B2; i += P2T)
                if (sm.cur[i] - i * cap2 > cap2) *sm.flag = 1;
        }
        ktd::lds_barrier();
        if (!failed && *sm.flag == 0) {
            for (uint32_t i = tid; i < B2; i += P2T) {
                fstart[(uint64_t)jl * B2 + i] = lo + (uint64_t)i * cap2;
                fend[(uint64_t)jl * B2 + i] = lo + sm.cur[i];
            }
        } else if (tid == 0) {
            fail[jl] = 1u;  // redone by part2_kernel<K, false, BIG> (exact fine boundaries) in the launch behind this one
            atomicAdd(&fail[p.B1], 1u);  // (how many of the job's buckets: finish_typed reports it, kt_shard.hip sizes its next job by it)
        }
        ktd::lds_barrier();
    }
}

// ---- part2 with write combining in LDS: whole, aligned cache lines only (round 4) -----------------------------------
// tools/ubench/scatter_runs.hip: 1024 output streams per workgroup, 8 GB of 8-byte keys appended in 128-byte runs - lines
// that are aligned are written at 4.1 TB/s, runs that start anywhere at 1.5 TB/s (64-byte aligned: 2.6; any alignment
// below 64 bytes: the same 1.5), and with half the compute units masked off level 2 ran as fast as with all of them: the
// pass was never bound by its LDS work but by the partial 64-byte granules at both ends of every run it wrote.  So here
// nothing but whole lines is written.  There is no sort buffer: the LDS holds ONE LINE PER FINE BUCKET (B2 x 128 bytes), a
// key's place in its bucket's stream comes from one returning LDS atomic (q = fill[d]++), and the chunk is emitted in
// GENERATIONS: generation g = the keys with q / LK == g (LK = keys per line) are written to their slots of the bucket's
// line, barrier, every bucket whose stream has reached (g + 1) * LK writes its line - 8 lanes x 16 bytes, so a wave's store
// instruction is eight whole lines - barrier.  A chunk of 8 K keys over 1024 buckets (half a line per bucket with 64-bit
// keys, a quarter with 32-bit ones) takes one or two flushing generations and a last one that only writes (what is left in
// the lines is carried into the next chunk).  What a bucket
// still holds when its input ends goes out as one partial line.
// Same contract as part2_fast_kernel: fixed fine regions (cap2 keys each, a multiple of LK so that every region starts on
// a line), a bucket that outgrows one raises fail[jl] and is redone by the general kernel.
template <class K>
struct SwwcShared {
    K *buf;             // [B2][LK] the buckets' lines
    uint32_t *fc;       // [B2][2] {fill: keys in the bucket's stream since its last line boundary; cur: keys written out}
    uint32_t *flags;    // [0], [1]: some bucket has a line to write in generation g (g & 1); [2]: a bucket is outgrowing its room
    static size_t bytes(uint32_t B2) { return (size_t)B2 * 128 + (size_t)B2 * 8 + 64; }
    __device__ SwwcShared(unsigned char *raw, uint32_t B2) {
        buf = reinterpret_cast<K *>(raw);
        fc = reinterpret_cast<uint32_t *>(raw + (size_t)B2 * 128);
        flags = fc + 2 * B2;
    }
};
// threads of a workgroup and keys of a thread per chunk: 1024 x 8 with 64-bit keys (half a line per fine bucket and
// chunk; 512 x 32: 10.6 ms, 768 x 16: 10.0 ms, 1024 x 8: 9.5 ms at ctr k=31 - 1024 x 16 does not fit the 128 registers
// of a 16-wave workgroup), 1024 x 8 with 32-bit keys too (ctr k=15: 512 x 32 13.9 ms, 1024 x 16 11.5-11.7, 1024 x 8 with four
// buckets per flush trip 11.1-11.3)
#ifndef KT_SWWC_T64
#define KT_SWWC_T64 1024
#endif
#ifndef KT_SWWC_T32
#define KT_SWWC_T32 1024
#endif
template <class K>
constexpr int swwc_t() { return sizeof(K) == 8 ? KT_SWWC_T64 : KT_SWWC_T32; }

template <class K>
__global__ __launch_bounds__((swwc_t<K>()), (swwc_t<K>() / 256)) void part2_swwc_kernel(P2In in, Plan p, K *__restrict__ keys2,
                                                              uint64_t *__restrict__ fstart, uint64_t *__restrict__ fend,
                                                              uint32_t *__restrict__ fail) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const SwwcShared<K> sm(smem_raw, p.B2);
    constexpr K EMPTY = empty_of<K>();
#ifndef KT_SWWC_PER32
#define KT_SWWC_PER32 16  // keys of a 512-thread lane per chunk with 32-bit keys (1024 threads: half of it): 8 K keys = a quarter of
#endif                    // a line per fine bucket and chunk.  (At 512 threads: 32 -> 14.2 ms at k=15, 48 -> 16.2, 64 does not fit 256
                          // registers; at 1024 threads 32 -> 11.5 ms, 16 with four buckets per flush trip -> 11.1-11.2)
#ifndef KT_SWWC_PER64
#define KT_SWWC_PER64 8
#endif
    constexpr int P2T = swwc_t<K>(), PER = sizeof(K) == 8 ? KT_SWWC_PER64 : KT_SWWC_PER32 * 512 / P2T;
    constexpr uint32_t CH = (uint32_t)P2T * PER;       // 8192 keys per chunk, held in registers
    constexpr uint32_t LK = 128 / sizeof(K), LSH = sizeof(K) == 8 ? 4 : 5;  // keys per line
    constexpr uint32_t GL = 8;                          // lanes that write one line (16 bytes each)
    constexpr uint32_t KPL = LK / GL;                   // keys per lane of a line: 2 (4)
    const uint32_t tid = threadIdx.x;
    const uint32_t nd = p.B1;
    const uint32_t B2 = p.B2, cap2 = (uint32_t)p.cap2, dshift = 32 - p.b1 - p.b2;  // (in the hash's high word: digit2h)
    auto digit = [&](K stored) -> uint32_t { return ((uint32_t)(hash_of_stored<K>(stored, p) >> 32) >> dshift) & (B2 - 1u); };
    const uint32_t grp = tid / GL, gl = tid % GL;       // the line group this thread belongs to, its lane in it
    for (uint32_t jl = blockIdx.x; jl < nd; jl += gridDim.x) {
        const uint32_t n_seg = in.n_src;
        auto segment = [&](uint32_t sidx, const K *&base, uint64_t &n) {
            const kt_seg_src &q = in.srcs[sidx];
            const uint64_t c = q.counts[jl];
            base = reinterpret_cast<const K *>(q.keys) + (uint64_t)jl * q.cap1;
            n = c < q.cap1 ? c : q.cap1;
        };
        uint64_t total = 0;
        for (uint32_t sidx = 0; sidx < n_seg; sidx++) {
            const K *b;
            uint64_t n;
            segment(sidx, b, n);
            total += n;
        }
        if (fail[jl] != 0) continue;  // (marked already: the exact pass takes it)
        const uint64_t lo = in.out_of(jl, p.room1);
        for (uint32_t i = tid; i < 2 * B2; i += P2T) sm.fc[i] = 0;
        if (tid < 4) sm.flags[tid] = 0;
        ktd::lds_barrier();
        const kt_i32x4 outrs = buf_rsrc(keys2 + lo, (uint32_t)(p.room1 * sizeof(K)));
        // the running check: a fine bucket's fill against its room scaled to the keys seen so far, 7 % off, + 6 sigma + 32
        const uint32_t allow_a = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)__float_as_uint(0.93f * (float)cap2 / (float)(total ? total : 1)));
        const uint32_t allow_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(6.f * sqrtf((float)cap2) + 32.f));
        uint64_t seen = 0;
        bool failed = false;
        K kcur[PER], knext[PER];
        uint32_t qa[PER];  // generation << 16 | slot of the key in buf (d * LK + q % LK); ~0: no key
        // one line per group per trip: generation g's lines.  Returns nothing; raises flags[(g + 1) & 1] when some bucket
        // has yet another line, flags[2] when a bucket is outgrowing its room.
#ifndef KT_SWWC_FB
#define KT_SWWC_FB 4
#endif
        auto flush = [&](uint32_t g, float allowed) {
            // FB trips at a time: the buckets' states are read together, then the lines, then the stores go out (512 threads x 32
            // keys: FB = 2 10.4 -> 9.5 ms at k=31, FB = 4 did not fit the registers - and a spill in this kernel is not a
            // slowdown but a fault: see csrc/Makefile's checks.  At 1024 x 8 keys four fit: nothing at k=31 (9.3-9.6 either
            // way), 11.5 -> 11.1-11.2 ms at k=15 together with the shorter chunk)
            constexpr uint32_t FB = KT_SWWC_FB, NGR = P2T / GL;
            for (uint32_t d0 = 0; d0 < B2; d0 += NGR * FB) {
                uint32_t fl[FB], cu[FB];
                bool some = false;
#pragma unroll
                for (uint32_t b = 0; b < FB; b++) {
                    const uint32_t d = d0 + b * NGR + grp;
                    fl[b] = d < B2 ? sm.fc[2 * d] : 0u;
                    cu[b] = d < B2 ? sm.fc[2 * d + 1] : 0u;
                }
#pragma unroll
                for (uint32_t b = 0; b < FB; b++) some |= (fl[b] >> LSH) != 0;
                typedef uint32_t raw4 __attribute__((ext_vector_type(4)));
                raw4 v[FB];
#pragma unroll
                for (uint32_t b = 0; b < FB; b++) v[b] = raw4{0u, 0u, 0u, 0u};
                if (__builtin_amdgcn_ballot_w64(some) != 0) {
#pragma unroll
                    for (uint32_t b = 0; b < FB; b++) {
                        const uint32_t d = d0 + b * NGR + grp;
                        v[b] = *reinterpret_cast<const raw4 *>(&sm.buf[(size_t)(d < B2 ? d : 0) * LK + gl * KPL]);
                    }
                }
#pragma unroll
                for (uint32_t b = 0; b < FB; b++) {
                    const uint32_t d = d0 + b * NGR + grp;
                    if (d0 + b * NGR >= B2) break;  // (uniform: B2 is not a multiple of the batch)
                    const uint32_t fill = fl[b], cur = cu[b];
                    const bool go = (fill >> LSH) != 0;
                    const bool fits = cur + LK <= cap2;
                    const uint32_t off = go && fits ? (d * cap2 + cur + gl * KPL) * (uint32_t)sizeof(K) : BUF_DROP;
                    // (every trip issues its store - one that has nothing to write is dropped by the hardware - so that the number
                    // of stores per generation is fixed: the wait for the next chunk's loads counts them)
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v[b]), "v"(off), "s"(outrs));  // (s_nop: see buf_store_async)
                    if (go && gl == 0) {
                        // the bucket's stream moves on by a line: fill counts from the new boundary (the keys of the later
                        // generations have their places already: they were taken from fill before the first barrier)
                        sm.fc[2 * d] = fill - LK;
                        sm.fc[2 * d + 1] = cur + LK;
                        if ((fill >> LSH) > 1) sm.flags[(g + 1) & 1] = 1;
                        if (!fits || (float)(cur + fill) > allowed) sm.flags[2] = 1;
                    }
                }
            }
        };
        for (uint32_t sidx = 0; sidx < n_seg && !failed; sidx++) {
            const K *base;
            uint64_t n;
            segment(sidx, base, n);
            if (n == 0) continue;
            auto load_chunk = [&](uint64_t c0, K (&dst)[PER]) {  // thread t takes keys c0 + u * P2T + t; in flight until buf_wait
                const uint32_t left = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n - c0 > CH ? CH : (uint32_t)(n - c0)));
                const kt_i32x4 rs = buf_rsrc(base + c0, left * (uint32_t)sizeof(K));
                uint32_t voff = tid * (uint32_t)sizeof(K);
#pragma unroll
                for (int u = 0; u < PER; u += 8) buf_load8_async<K, P2T * (int)sizeof(K)>(&dst[u], rs, voff);
            };
            load_chunk(0, knext);
            buf_take<0>(kcur, knext);
            for (uint64_t c0 = 0; c0 < n; c0 += CH) {
                uint32_t tl = tid;  // (opaque per chunk: nothing derived from the thread index is carried across the loop)
                asm volatile("" : "+v"(tl));
                const uint32_t left = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n - c0 > CH ? CH : (uint32_t)(n - c0)));
                if (left != CH) {  // the segment's last chunk: what lies past its end was read as 0
#pragma unroll
                    for (int u = 0; u < PER; u++) kcur[u] = (uint32_t)u * P2T + tl < left ? kcur[u] : EMPTY;
                }
                const bool more = c0 + CH < n;
                if (more) load_chunk(c0 + CH, knext);  // the next chunk travels while this one is emitted
                // every key takes its place in its bucket's stream (an empty key - a page gap - takes none)
                if (tl == 0) sm.flags[0] = 0;  // (read last behind a barrier of the previous chunk, or as 0; set again in generation 0)
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t d = digit(kcur[u]);
                    const bool valid = kcur[u] != EMPTY;
                    const uint32_t q = atomicAdd(&sm.fc[2 * d], valid ? 1u : 0u);
                    qa[u] = valid ? ((q >> LSH) << 16) | (d * LK + (q & (LK - 1u))) : ~0u;
                }
                ktd::lds_barrier();
                if (sm.flags[2] != 0) {  // (raised by a flush before the barrier above: the same for every thread)
                    failed = true;
                    // the next chunk's loads are in flight into knext: they land before the loop is left - with fewer workgroups
                    // than buckets (KT_P2_GRID) the kernel goes on to its next bucket and loads into the same registers
                    if (more) buf_drain(knext);
                    break;
                }
                const uint64_t sn = seen + c0 + CH;
                const float allowed = (float)(uint32_t)(sn < total ? sn : total) * __uint_as_float(allow_a) + __uint_as_float(allow_b);
                // generation 0: the keys that fit the lines as they stand; does any bucket reach a line?
#pragma unroll
                for (int u = 0; u < PER; u++)
                    if ((qa[u] >> 16) == 0u) sm.buf[qa[u] & 0xFFFFu] = kcur[u];
                for (uint32_t i = tl; i < B2; i += P2T)
                    if ((sm.fc[2 * i] >> LSH) != 0) sm.flags[0] = 1;
                if (tl == 0) sm.flags[1] = 0;  // (flush(0) raises it; its last readers are a barrier behind)
                ktd::lds_barrier();
                bool any = sm.flags[0] != 0;
                uint32_t gens = 0;  // flushing generations of this chunk: P2T / GL... B2 / (P2T / GL) stores each
                for (uint32_t g = 0; any; g++) {
                    flush(g, allowed);
                    gens++;
                    ktd::lds_barrier();
                    any = sm.flags[(g + 1) & 1] != 0;
                    // generation g + 1: into the lines that have just gone out.  (flags[g & 1], which flush(g + 1) raises, was
                    // read by everybody before flush(g): cleared here, behind the barrier that followed it)
                    if (tl == 0) sm.flags[g & 1] = 0;
#pragma unroll
                    for (int u = 0; u < PER; u++)
                        if ((qa[u] >> 16) == g + 1u) sm.buf[qa[u] & 0xFFFFu] = kcur[u];
                    if (any) ktd::lds_barrier();  // (the last generation only writes: the next chunk's count may follow at once)
                }
                if (more) {
                    // the next chunk's keys are in (they were requested before the count); the line stores issued since - a
                    // fixed number per generation - stay in flight
                    constexpr int SPG = (1024 + P2T / GL - 1) / (P2T / GL);  // stores per generation at B2 = 1024
                    const uint32_t issued = gens * ((B2 + P2T / GL - 1) / (P2T / GL));
                    if (issued == SPG) buf_take<SPG>(kcur, knext);
                    else if (issued == 2 * SPG) buf_take<2 * SPG>(kcur, knext);
                    else if (issued == 3 * SPG) buf_take<3 * SPG>(kcur, knext);
                    else buf_take<0>(kcur, knext);
                }
            }
            seen += n;
        }
        ktd::lds_barrier();
        // what the lines still hold goes out as partial lines; the fine buckets' extents
        if (!failed) {
            for (uint32_t d0 = 0; d0 < B2; d0 += P2T / GL) {
                const uint32_t d = d0 + grp;
                if (d >= B2) continue;
                const uint32_t fill = sm.fc[2 * d], cur = sm.fc[2 * d + 1];
                if (cur + fill > cap2) {
                    sm.flags[2] = 1;
                    continue;
                }
                for (uint32_t e = gl * KPL; e < gl * KPL + KPL; e++)
                    if (e < fill) keys2[lo + (uint64_t)d * cap2 + cur + e] = sm.buf[(size_t)d * LK + e];
                if (gl == 0) {
                    fstart[(uint64_t)jl * B2 + d] = lo + (uint64_t)d * cap2;
                    fend[(uint64_t)jl * B2 + d] = lo + (uint64_t)d * cap2 + cur + fill;
                }
            }
        }
        ktd::lds_barrier();
        if ((failed || sm.flags[2] != 0) && tid == 0) {
            fail[jl] = 1u;  // redone by part2_kernel<K, false, BIG> in the launch behind this one
            atomicAdd(&fail[p.B1], 1u);
        }
        ktd::lds_barrier();
    }
}

// ---- build: one workgroup per fine bucket = one range of the table ---------------------------------------
// A range (kt_table.hpp) is a closed, circular linear-probing table of RS = 1024 * m8 slots, so its image fits LDS
// (<= 96 KB of keys + counts): the bucket's keys - on top of the range's current image when the table already holds
// data - are inserted with LDS atomics and the image is written out with coalesced non-temporal 16-byte stores (empty
// slots included, so an empty table needs no clear).  Nothing ever leaves its range: no clean-up pass through the
// global atomic path, and an existing range can be rebuilt in place (later batches merge without global atomics).
// More distinct keys than slots in a range (a table that is too small) go to the spill list and fail there, loudly.
//
// Measured this round on the way here (ctr k=31, 25 M reads, build kernel alone; profiles/r2_build_sweep.txt):
//  * the insert is not bound by LDS atomic throughput (tools/ubench/lds_atomics.hip: 2.9 random 64-bit CAS per clock
//    and CU; the kernel needs a tenth of that) but by the longest probe walk of a workgroup: at load 0.8 a range's
//    longest run of taken slots is a few hundred, and one lane walks it a slot per LDS round trip while the
//    workgroup waits at the barrier.  Load 0.47 / 0.56 / 0.70 / 0.80: 24.9 / 27.8 / 38.0 / 61.6 ms.
//  * several insert machines per lane (more CAS in flight) made it worse (0.8: 61.6 -> 84.7 -> 116.8 ms for 1 / 2 / 4):
//    fewer keys per machine, same longest walk.  Handing finished lanes the wave's next keys (ballot + rank +
//    ds_bpermute) and reading before the CAS: 53.8 ms.  A wave-wide 64-slot scan for walks past 16 slots: 40.2 ms.
//  * deduplicating in a sparser LDS table and computing the probing layout with a counting sort + prefix maximum
//    instead of probing (load-independent): 39.6 ms - dedupe 20, placement 10, image 6, none of it overlapping.
//  * so the table keeps a load factor near 0.5, where the plain state machine below is bound by the image it writes
//    (103 GB at 5 TB/s), and the image is what a smaller-slot layout would have to shrink.
//  * dense state, end of the round: giving the four prefetched keys of a lane their first probe together (four CAS in
//    flight, the collided ones queued ahead of the rest) made the kernel slower again, 20.8 against 18.8 ms; reading
//    2 or 4 of the pack phase's 64-slot passes before writing any back changed nothing (18.7 ms).  The machine written
//    with selects instead of branches (the branchy loop spends ~25 scalar instructions per trip on exec-mask
//    bookkeeping, and the CU's scalar unit is busy half of the kernel's time): 27.6 against 18.5 ms.
static_assert(LOG2_S == kttab::LOG2_RANGE, "a fine bucket is a range of the table");

constexpr int BUILD_T = KT_BUILD_T;
// claim lists of a dense build: the slot of every key a wave has placed first, in the order it placed them - three keys per
// lane at most (a range of up to 3072 keys; ranges of hashed distinct keys hold ~2900)
constexpr uint32_t LIST_CAP = 192, LIST_KEYS = LIST_CAP * (BUILD_T / 64);

template <class K>
struct lds_word;
template <>
struct lds_word<uint64_t> { using type = unsigned long long; };
template <>
struct lds_word<uint32_t> { using type = unsigned int; };

// DENSE (a fresh table only): instead of the range's image, its occupied slots alone are written, packed at the front
// of the range's slots, and their number goes to range_counts[range].  That is all kt_ctr_size / kt_ctr_export need
// (the usual fate of a table: counted once, written out), and it is 48 GB instead of 103 GB at k=31 / 25 M reads; the
// probing image is produced from it in place (materialize_kernel) the first time something has to probe.
// EXT (dense only): the packed entries do not go to the range's own slots but straight into the caller's export
// arrays (kt_ctr_export_target) as (key, occurrences) - the build's output IS the export, and the pass that copied
// 36 GB of packed ranges into the export arrays (ctr k=31: 12-14 ms of a 65 ms step) is gone.
// Where a range's entries go cannot be a prefix sum (how many distinct keys a range has is known only once it has been
// built), and a global cursor does not work either: a returning atomic on ONE address is served at 83 M/s
// (tools/ubench/same_addr_atomic.hip), and the compiler waits for it where it is issued, so each one holds a
// 1024-thread workgroup for a memory round trip under load (one per range: build 17.9 -> 32 ms; one per block of 8192
// entries, requested ahead: 20.1 ms).  So there is no cursor: output space is cut into BLOCKS of XBLK entries and
// workgroup b of G owns blocks b, b + G, b + 2 G, ...; it packs its ranges back to back into them (a range that does
// not fit the rest of a block continues at the front of the workgroup's next one).  The hash spreads the keys, so
// every workgroup fills about the same number of blocks; what is left are holes - the unused tail of every
// workgroup's last block, and whole blocks of workgroups that needed one fewer than the others - about 0.5 % of the
// entries.  ext_scan_kernel / ext_patch_kernel then move the entries that lie beyond the packed length n into the
// holes below n, and the arrays hold exactly n entries in [0, n).  The order of the entries is deterministic (the
// contract says unspecified, like the reference's map scan).
// Positions are "virtual": [0, max) is the caller's arrays, [max, max + ovf_cap) the library's scratch behind them,
// because blocks + holes reach past n even when n <= max (the caller's arrays may be exactly n long).  Keys that are
// NOT spread evenly over the workgroups (an adversarial input) can push the extent past the scratch: ext_scan_kernel
// raises flag 4 and the host builds the ranges again the ordinary way (finish_typed).
// The table keeps no per-range record of where entries went: whatever needs the probing image afterwards re-inserts
// the exported pairs (kt_table_image).
constexpr uint32_t XBLK = 8192;  // >= the slots of a range: a range spans at most two blocks
static_assert(XBLK >= (1u << LOG2_S), "a range's entries fit one block");
struct ExtOut {
    uint64_t *keys;
    uint32_t *counts;
    uint64_t max;           // entries of the caller's arrays
    uint64_t *ovf_keys;     // scratch: virtual positions max ... max + ovf_cap
    uint32_t *ovf_counts;
    uint64_t ovf_cap;
    uint64_t *extent;       // virtual positions in use: max over the workgroups of the end of their last block
    uint32_t *fill;         // [max_blocks]: entries at the front of a block (zeroed before the build)
    uint64_t max_blocks;    // blocks fill[] has room for (an extent that runs away is cut off there: ext_scan_kernel reports it)
    __device__ __forceinline__ void put(uint64_t pos, uint64_t key, uint32_t occ) const {
        if (pos < max) {
            __builtin_nontemporal_store(key, keys + pos);
            __builtin_nontemporal_store(occ, counts + pos);
        } else if (pos - max < ovf_cap) {
            ovf_keys[pos - max] = key;
            ovf_counts[pos - max] = occ;
        }  // (else: the extent is past the scratch - ext_scan_kernel reports it, the build is redone)
    }
    __device__ __forceinline__ void get(uint64_t pos, uint64_t &key, uint32_t &occ) const {
        if (pos < max) {
            key = keys[pos];
            occ = counts[pos];
        } else {
            key = ovf_keys[pos - max];
            occ = ovf_counts[pos - max];
        }
    }
};

// Two 16-wave workgroups per CU = 8 waves per SIMD, which the hardware grants only to kernels of at most 64 VGPRs AND at
// most 80 SGPRs (MI355X_MICROARCH.md, "Residency": 82-96 SGPRs -> 7 waves per SIMD, whatever the occupancy API says).
// The EXT variant first compiled to 82-85 SGPRs - one workgroup per CU, 32 instead of 18 ms, with every other
// suspect (the cursor atomics, the stores, the claim counting) measured innocent one by one - hence the cap.
#ifndef KT_BUILD_DHASH
#define KT_BUILD_DHASH 1   // 0: dense builds probe linearly too (A/B builds)
#endif
#ifndef KT_BUILD_SGPRS
#define KT_BUILD_SGPRS 80
#endif
// DIRECT (32-bit keys, a fresh dense build of a table with exactly 4^k slots, i.e. n = 2k hash bits: plan_job): the
// table's hash is a bijection of the k-mers onto the slots (ktd::nhash), so a range's image is its COUNTERS alone - a key
// is one LDS add at the low 13 bits of its hash, no key stored, no compare, no probe chain, no lane waiting for another
// lane's chain - and the entries come out of the counters that are not zero: key = nhash_inv(range's bits | index).
// (the reference's `table[kmer] += 1`, counter/src/lib.rs:126-130)
template <class K, bool MERGE, bool DENSE, bool EXT = false, bool DIRECT = false>
__global__ __launch_bounds__(BUILD_T) __attribute__((amdgpu_num_sgpr(KT_BUILD_SGPRS), amdgpu_waves_per_eu(8, 8))) void build_kernel(const K *__restrict__ keys2,
                                                        const uint64_t *__restrict__ fstart,
                                                        const uint64_t *__restrict__ fend, Plan p,
                                                        Slot *__restrict__ slots, uint64_t *__restrict__ spill_n,
                                                        uint64_t *__restrict__ spill_keys,
                                                        uint32_t *__restrict__ spill_counts, uint64_t spill_cap,
                                                        uint32_t *__restrict__ spill_ovf,
                                                        uint64_t *__restrict__ distinct,
                                                        uint32_t *__restrict__ range_counts, ExtOut xo) {
    static_assert(!(MERGE && DENSE), "a dense build starts from an empty table");
    static_assert(!EXT || DENSE, "only a dense build writes to the export arrays");
    static_assert(!DIRECT || (DENSE && sizeof(K) == 4), "the direct build: 32-bit keys, dense");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using W = typename lds_word<K>::type;
    constexpr K EMPTY = empty_of<K>();
    const uint32_t RS = p.m8 << (LOG2_S - 3);
    K *const skeys = reinterpret_cast<K *>(smem_raw);
    uint32_t *const scounts = reinterpret_cast<uint32_t *>(smem_raw + (size_t)RS * sizeof(K));
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint64_t n_fine = (uint64_t)(p.B1) * p.B2;  // the ranges this table holds
    const uint32_t shift = 64 - p.n;
    // (the image in LDS holds keys in their stored form, like the key arrays: to_stored / from_stored; a key's home
    // position inside the range is kttab::probe_of's: home_w below)
    auto spill = [&](uint64_t key, uint32_t occurrences) {
        if (DENSE) {  // (no image to probe afterwards: a full range is reported at once)
            atomicOr(spill_ovf, 1u);
            return;
        }
        const uint64_t at = atomicAdd(reinterpret_cast<unsigned long long *>(spill_n), 1ull);
        if (at < spill_cap) {
            spill_keys[at] = key;
            spill_counts[at] = occurrences;
        } else {
            atomicOr(spill_ovf, 1u);
        }
    };
    __shared__ uint32_t runs[BUILD_T / 64];  // DENSE: entries packed by each wave
    // DENSE with claim lists (p.lists: the launch gave the room): a wave notes the slot of every key it places first, so
    // the occupied slots need not be looked for afterwards - no pack pass over the whole image, no clearing pass either
    // (the copy-out empties the slots it has read), one barrier fewer.  All three were proportional to the range's SLOTS
    // (6144 at load 0.47), the list is proportional to its keys.  A range of more than LIST_KEYS keys takes the pack path.
    uint16_t *const mylist = reinterpret_cast<uint16_t *>(smem_raw + (size_t)RS * (sizeof(K) + 4)) + (threadIdx.x >> 6) * LIST_CAP;
    bool dirty = true;  // the image holds something: cleared whole before the next insert (workgroup uniform)
    // EXT: the current block [xpos, xend) and how many blocks this workgroup has taken.  Every thread carries the same
    // (workgroup-uniform) values: a range's entry count D comes out of the pack's scan in every wave, so where its entries
    // go - the rest of the current block, then the front of the workgroup's next one - needs no shared state at all.
    uint64_t xpos = 0, xend = 0;
    uint32_t xused = 0;
    long long placed = 0;  // per thread: occupied slots written - occupied slots found (MERGE)
    // A range starts with two dependent global reads (its bounds, then its first keys): ~4 us during which the
    // workgroup would do nothing, 1500 times over.  Both are taken one range ahead: the next range's bounds are
    // requested before this range's insert, its first four keys per lane before this range's copy-out.
    uint64_t lo = 0, hi = 0;
    K head[4] = {EMPTY, EMPTY, EMPTY, EMPTY};
    auto load_head = [&](uint64_t l, uint64_t h, K (&dst)[4]) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t i = l + tid + (uint64_t)u * BUILD_T;
            dst[u] = i < h ? keys2[i] : EMPTY;
        }
    };
    if (blockIdx.x < n_fine) {
        lo = fstart[blockIdx.x];
        hi = fend[blockIdx.x];
        load_head(lo, hi, head);
    }
#if KT_ABLATION
    unsigned long long tph = __builtin_readcyclecounter();
    uint32_t phs[8] = {};
#endif
    for (uint64_t fb = blockIdx.x; fb < n_fine;) {
        const uint64_t nfb = fb + gridDim.x;
        uint64_t nlo = 0, nhi = 0;
        if (nfb < n_fine) {
            nlo = fstart[nfb];
            nhi = fend[nfb];
        }
        if (MERGE && lo == hi) {  // nothing new for this range: it stays as it is
            fb = nfb;
            lo = nlo;
            hi = nhi;
            load_head(lo, hi, head);
            continue;
        }
        if (MERGE) {
            // what the range holds already goes back where it is: the old image is a valid probing layout
            for (uint32_t i = tid; i < RS; i += BUILD_T) {
                const uint4 v = reinterpret_cast<const uint4 *>(slots + fb * RS)[i];
                const bool occ = (v.x & v.y) != 0xFFFFFFFFu;
                skeys[i] = occ ? to_stored<K>(((uint64_t)v.y << 32) | v.x) : EMPTY;
                scounts[i] = v.z;
                placed -= occ;
            }
        } else if (dirty) {
            for (uint32_t i = tid; i < RS; i += BUILD_T) {
                if (!DIRECT) skeys[i] = EMPTY;
                scounts[i] = 0;
            }
        }
        const bool listed = !DIRECT && DENSE && p.lists != 0 && hi - lo <= LIST_KEYS && hi - lo <= RS;  // (workgroup uniform)
        KT_PH(0);
        if (MERGE || dirty) ktd::lds_barrier();  // (a listed range leaves the image empty behind its end barrier)
        KT_PH(1);
        uint32_t wc = 0;  // DENSE: entries of this wave - listed: the claims of its lanes; else: counted by the pack pass
        uint32_t m0 = ~0u, m1 = ~0u;  // listed: the slots this lane has claimed (all ones: none)
        if constexpr (DIRECT) {
            // every key of the range: one LDS add at the low 13 bits of its hash (RS = 8192: m8 = 8).  Four keys per lane in
            // flight, the first four requested a range ahead (head[])
            const uint32_t sh32 = 32u - p.kbits;  // (hash_of_stored's word >> (64 - n), n = kbits: the hash itself)
            auto bump = [&](K key) {
                if (key != EMPTY) atomicAdd(&scounts[ktd::nhash((uint32_t)key, p.kbits) & (S - 1u)], 1u);
            };
            (void)sh32;
            uint64_t bbase = lo;
            for (;;) {
                const bool more = bbase + 4ull * BUILD_T < hi;  // (workgroup uniform)
                K nxt[4] = {EMPTY, EMPTY, EMPTY, EMPTY};
                if (more) load_head(bbase + 4ull * BUILD_T, hi, nxt);
#pragma unroll
                for (int u = 0; u < 4; u++) bump(head[u]);
                if (!more) break;
                bbase += 4ull * BUILD_T;
#pragma unroll
                for (int u = 0; u < 4; u++) head[u] = nxt[u];
            }
        } else {
            // Every lane runs its own insert state machine over its keys (lo + tid, + BUILD_T, ...): one probe per
            // trip (the CAS itself reports what the slot holds), and a lane that has placed its key moves on to its
            // next one at once.  ("For each key: probe until placed" makes the wave wait for its longest probe chain on
            // every key.)  The keys come in BATCHES of four per lane, loaded together: a range of hashed distinct keys is
            // one batch (~2900 keys for 1024 lanes) and its keys are the ones requested a range ahead; a longer range (a
            // repeat of the genome, a skewed batch) takes further batches, each requested before the one in front of it
            // is inserted.  (Round 3's loop refilled a four-deep queue one key at a time: a compare, a branch, an
            // address and - because the load sat in the loop - a wait for ALL the workgroup's stores in flight on every
            // key placed, for a refill that hashed keys never need.)
            // A dense build's image is scratch - only the packed entries leave the kernel - so it need not be the
            // table's probing layout: its collisions step by a key-dependent prime (coprime to every range size 1024 * m8)
            // instead of 1.  The insert phase ends when the slowest of the 1024 lanes has placed its keys, i.e. after the
            // LONGEST probe chain of the range, and linear probing's clusters make that chain ~3x longer at 0.47 load.
            // (64-bit keys only: the 32-bit keys of k <= 16 hash nearly collision-free - ctr k=15's build measured 11.5 ms
            // with linear probing and 11.8 with the strides, where k=31's went from 19.2 to 16.9 ms)
            constexpr bool STRIDES = KT_BUILD_DHASH && DENSE && sizeof(K) == 8;
            // the hash bits the insert looks at, in one 32-bit word: bits 3.. = position among the range's S homes, bits
            // 0..2 = which stride (one v_alignbit; the 64-bit shifts are not full-rate instructions)
            const uint32_t sh3 = shift - 3;  // (the bulk path takes tables of 2^15 .. 2^34 hash positions: shift is 30 .. 49)
            // (which of the two forms applies is the table's size: decided outside the loop, insert_batch's HI - left to
            // the compiler it is a scalar compare and two branches on every trip)
            auto hword = [&](K stored, auto hi_only) -> uint32_t {
                const uint64_t h = hash_of_stored<K>(stored, p);
                const uint32_t hi32 = (uint32_t)(h >> 32), lo32 = (uint32_t)h;
                if constexpr (decltype(hi_only)::value) return hi32 >> (sh3 - 32);
                else return __builtin_amdgcn_alignbit(hi32, lo32, sh3);
            };
            auto home_w = [&](uint32_t w) -> uint32_t { return __umul24((w >> 3) & (S - 1), p.m8) >> 3; };
            auto stride_w = [&](uint32_t w) -> uint32_t {
                if (!STRIDES) return 1u;
                // 11 13 17 19 23 29 31 37, picked by the word's low three bits: one byte permute of the table
                return __builtin_amdgcn_perm(0x251f1d17u, 0x13110d0bu, (w & 7u) | 0x0c0c0c00u);
            };
            // CHECK: a walk that has come round the whole range reports it as full.  A fresh range that receives no more keys
            // than it has slots cannot fill up, and its loop goes without the walk's counter - two vector and five scalar
            // instructions of a trip that issues ~45, in a loop bound by instruction issue (eight waves per SIMD going
            // round it: 17.2 against 19.2 ms for the kernel).
            auto insert_batch = [&](auto chk, auto lst, auto hi_only) {
                constexpr bool CHECK = decltype(chk)::value, LISTED = decltype(lst)::value;
                // (a listed range has at most three keys per lane: its queue is one key shorter)
                K cur = head[0], q0 = head[1], q1 = head[2], q2 = LISTED ? EMPTY : head[3];
#if KT_ABLATION
                if (p.dbg & 1u) cur = EMPTY;
#endif
                uint32_t w = hword(cur, hi_only);
                uint32_t s = home_w(w), step = stride_w(w), probes = 0;
                while (cur != EMPTY) {
                    const K v = (K)atomicCAS(reinterpret_cast<W *>(&skeys[s]), (W)EMPTY, (W)cur);
                    // claimed: first occurrence, stored count stays 0; dup: the key is there already (cur is not EMPTY
                    // here, so the two exclude each other - written as two flat tests because the nested form costs
                    // the loop half a dozen scalar instructions of mask bookkeeping per trip)
                    const bool claimed = v == EMPTY, dup = v == cur;
                    bool done = claimed | dup;
                    if (dup) atomicAdd(&scounts[s], 1u);
                    if constexpr (LISTED) {
                        // the lane remembers the slots it has claimed - three at most, 16 bits each, shifted in: two
                        // instructions of a loop that is bound by instruction issue.  (Round 4 appended them to the
                        // wave's list inside the loop: a vote, two bit counts, an address, a 16-bit write, a broadcast
                        // of the new length - a dozen instructions on every trip in which any lane claimed a slot.)
                        if (claimed) {
                            m1 = __builtin_amdgcn_alignbit(m1, m0, 16);
                            m0 = (m0 << 16) | s;
                        }
                    }
                    s += step;                     // round the range (step 1: kttab::Probe)
                    if (s >= RS) s -= RS;
                    if (CHECK && !done && ++probes >= RS) {  // the range is full: the table is too small
                        spill(from_stored<K>(cur), 1u);
                        done = true;
                    }
                    if (done) {
                        cur = q0;
                        q0 = q1;
                        q1 = q2;
                        q2 = EMPTY;
                        w = hword(cur, hi_only);
                        s = home_w(w);
                        step = stride_w(w);
                        if (CHECK) probes = 0;
                    }
                }
            };
            const bool roomy = !MERGE && hi - lo <= RS;  // (workgroup uniform)
            uint64_t bbase = lo;
            for (;;) {
                const bool more = bbase + 4ull * BUILD_T < hi;  // (workgroup uniform; false for hashed distinct keys)
                K nxt[4] = {EMPTY, EMPTY, EMPTY, EMPTY};
                if (more) load_head(bbase + 4ull * BUILD_T, hi, nxt);
                auto insert = [&](auto hi_only) {
                    if (listed) insert_batch(std::false_type{}, std::true_type{}, hi_only);
                    else if (roomy) insert_batch(std::false_type{}, std::false_type{}, hi_only);
                    else insert_batch(std::true_type{}, std::false_type{}, hi_only);
                };
                if (sh3 >= 32) insert(std::true_type{});
                else insert(std::false_type{});
                if (!more) break;
                bbase += 4ull * BUILD_T;
#pragma unroll
                for (int u = 0; u < 4; u++) head[u] = nxt[u];
            }
        }
        if (DENSE && listed) {
            // the wave's list = its lanes' claimed slots one lane after the other (one scan per range; the wave reads
            // its own list back below, and LDS executes a wave's accesses in order)
            const uint32_t c = (uint32_t)((m0 & 0xFFFFu) != 0xFFFFu) + (uint32_t)((m0 >> 16) != 0xFFFFu) +
                               (uint32_t)((m1 & 0xFFFFu) != 0xFFFFu);
            const uint32_t inc = wave_incl_scan(c);
            uint16_t *const at = mylist + (inc - c);
            if (c > 0) at[0] = (uint16_t)m0;
            if (c > 1) at[1] = (uint16_t)(m0 >> 16);
            if (c > 2) at[2] = (uint16_t)m1;
            wc = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            if (lane == 0) runs[tid >> 6] = wc;  // (known here: no pack pass, no barrier of its own)
        }
        KT_PH(2);
        ktd::lds_barrier();
        KT_PH(3);
        load_head(nlo, nhi, head);  // the next range's first keys travel while this range is written out
        uint4 *dst = reinterpret_cast<uint4 *>(slots + fb * RS);
        if (DENSE) {
            // The occupied slots, packed.  Every wave first packs its own sixteenth of the image in place (wave
            // synchronous: 64 slots into registers, ballot, back at the wave's cursor - entries only move towards the
            // front of the wave's share, so no other wave is involved), then the sixteen packed runs are copied out
            // one after the other.  (Writing each (wave, pass) run of <= 64 entries straight to global memory left
            // every store starting and ending inside cache lines: 36 GB took 14 ms.)
            constexpr uint32_t NW = BUILD_T / 64;
            const uint32_t wave = tid >> 6, share = RS / NW;  // RS = 1024 * m8: a multiple of 64 per wave
            // (with claim lists there is nothing to pack: the wave's entries are the slots on its list, wc of them)
            if (!listed) {
                for (uint32_t i0 = 0; i0 < share; i0 += 64) {
                    const uint32_t i = wave * share + i0 + lane;
                    K kk;
                    uint32_t cc;
                    if constexpr (DIRECT) {  // a counter that is not zero is an entry: its key from the slot's index
                        const uint32_t c1 = scounts[i];
                        cc = c1 - 1u;        // (stored: occurrences - 1, like the probing builds)
                        kk = c1 ? (K)ktd::nhash_inv(((uint32_t)fb << LOG2_S) | i, p.kbits) : EMPTY;
                    } else {
                        kk = skeys[i];
                        cc = scounts[i];
                    }
                    const uint64_t bal = __ballot(kk != EMPTY);
                    if (kk != EMPTY) {
                        const uint32_t at = wave * share + wc + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                        skeys[at] = kk;      // at <= i: behind or at what this wave has read (LDS executes a wave's
                        scounts[at] = cc;    // accesses in order)
                    }
                    wc += (uint32_t)__popcll(bal);
                }
                if (lane == 0) runs[wave] = wc;
                KT_PH(4);
                ktd::lds_barrier();
                KT_PH(5);
            }
            // entry e of this wave's run: from its packed share of the image, or from the slot its list names - which is
            // emptied behind the read (the next range finds the image clean)
            auto take = [&](uint32_t e, K &kk, uint32_t &cc) {
                const uint32_t src = listed ? (uint32_t)mylist[e] : wave * share + e;
                kk = skeys[src];
                cc = scounts[src];
                if (listed) {
                    skeys[src] = EMPTY;
                    scounts[src] = 0;
                }
            };
            // where this wave's run goes = the runs before it: an inclusive scan of the sixteen counts in lanes 0..15
            // (row_shr inside one DPP row)
            static_assert(NW == 16, "the wave counts are scanned inside one DPP row");
            uint32_t inc = lane < NW ? runs[lane] : 0u;
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, true);
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, true);
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, true);
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, true);
            const uint32_t D = (uint32_t)__builtin_amdgcn_readlane((int)inc, NW - 1);
            const uint32_t uw = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
            const uint32_t pre = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)uw) - wc;
            // a dense range is keys[D] at the front of the range's bytes and counts[D] from byte 8 * RS on: 12 bytes
            // per entry instead of a 16-byte slot (nothing probes this layout; materialize_kernel reads it back).
            // Every wave copies its own run: ~240 consecutive entries, with the lanes lined up on 32-entry
            // boundaries of the destination (256 bytes of keys, 128 of counts), so that only the two ends of a run
            // share cache lines with the neighbouring waves' runs.  (Finding the run of element e by a 16-way
            // compare chain, so that thread t could write element t, t + 1024, ..., took a third of the kernel's
            // VALU instructions.)
            if constexpr (EXT) {
                // where the range's D entries go: b1 + i for i < l1, then b2 + (i - l1)
                const uint64_t b1 = xpos;
                uint64_t b2 = 0;
                uint32_t l1 = D;
                if (xpos + D <= xend) {
                    xpos += D;
                } else {
                    l1 = (uint32_t)(xend - xpos);
                    if (tid == 0 && xend && xend / XBLK - 1 < xo.max_blocks) xo.fill[xend / XBLK - 1] = XBLK;  // the block just finished is full
                    b2 = ((uint64_t)blockIdx.x + (uint64_t)xused * gridDim.x) * XBLK;
                    xused++;
                    xpos = b2 + (D - l1);
                    xend = b2 + XBLK;
                }
                // where this wave's run starts (the same for all its lanes), and whether the run is one plain stretch of
                // the caller's arrays - nearly always: then the copy is the ordinary one, from a wave-uniform base; a run
                // that straddles two blocks or reaches the scratch takes the entry-by-entry path
                uint64_t at0 = ktd::uniform64(pre < l1 ? b1 + pre : b2 + (pre - l1));
#if KT_ABLATION
                if (p.dbg & 0x40u) at0 &= ~31ull;  // (timing only: every wave's run starts on a line - overlapping its neighbour's)
#endif
                const uint32_t skew = (uint32_t)at0 & 31u;
                const bool plain = (pre >= l1 || pre + wc <= l1) && at0 + wc <= xo.max;
                if (plain) {
                    uint64_t *const dk = xo.keys + at0;
                    uint32_t *const dc = xo.counts + at0;
                    for (uint32_t j = lane; j < wc + skew; j += 64) {
#if KT_ABLATION
                        if ((p.dbg & 2u) && skeys[0] != (K)0x1234567u) continue;  // (timing only: the build without its stores)
#endif
                        if (j >= skew) {
                            K kk;
                            uint32_t cc;
                            take(j - skew, kk, cc);
                            __builtin_nontemporal_store(from_stored<K>(kk), dk + (j - skew));
                            __builtin_nontemporal_store(cc + 1u, dc + (j - skew));
                        }
                    }
                } else {
                    for (uint32_t j = lane; j < wc + skew; j += 64) {
                        if (j >= skew) {
                            const uint32_t i = pre + (j - skew);
                            K kk;
                            uint32_t cc;
                            take(j - skew, kk, cc);
                            xo.put(i < l1 ? b1 + i : b2 + (i - l1), from_stored<K>(kk), cc + 1u);
                        }
                    }
                }
                if (tid == 0) placed += D;
            } else {
                uint64_t *const dkeys = reinterpret_cast<uint64_t *>(dst);
                uint32_t *const dcounts = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(dst) + (size_t)RS * 8);
                const uint32_t skew = pre & 31u;
                for (uint32_t j = lane; j < wc + skew; j += 64) {
                    if (j >= skew) {
                        K kk;
                        uint32_t cc;
                        take(j - skew, kk, cc);
                        __builtin_nontemporal_store(from_stored<K>(kk), dkeys + pre + (j - skew));
                        __builtin_nontemporal_store(cc, dcounts + pre + (j - skew));
                    }
                }
                if (tid == 0) {
                    range_counts[fb] = D;
                    placed += D;
                }
            }
#if KT_ABLATION
        } else if (!(p.dbg & 8u)) {
#else
        } else {
#endif
            for (uint32_t i = tid; i < RS; i += BUILD_T) {
                const K kk = skeys[i];
                const uint64_t key = kk == EMPTY ? KT_EMPTY_KEY : from_stored<K>(kk);
                placed += kk != EMPTY;
#if KT_BUILD_NT
                typedef uint32_t raw4 __attribute__((ext_vector_type(4)));
                const raw4 raw = {(uint32_t)key, (uint32_t)(key >> 32), scounts[i], 0u};
                __builtin_nontemporal_store(raw, reinterpret_cast<raw4 *>(dst + i));
#else
                dst[i] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), scounts[i], 0u);
#endif
            }
        }
        dirty = !listed;
        KT_PH(6);
        ktd::lds_barrier();
        KT_PH(7);
        fb = nfb;
        lo = nlo;
        hi = nhi;
    }
#if KT_ABLATION
    if (tid == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&kt_dbg_phase[i], (unsigned long long)phs[i]);
#endif
    if (EXT && tid == 0 && xend) {  // the last block is partly filled; the extent of the output is the furthest block's end
        if (xend / XBLK - 1 < xo.max_blocks) xo.fill[xend / XBLK - 1] = XBLK - (uint32_t)(xend - xpos);
        atomicMax(reinterpret_cast<unsigned long long *>(xo.extent), (unsigned long long)xend);
    }
    // the table's distinct counter: one atomic per wave
    for (int o = 32; o > 0; o >>= 1) placed += __shfl_down(placed, o, 64);
    if (lane == 0 && placed) atomicAdd(reinterpret_cast<unsigned long long *>(distinct), (unsigned long long)placed);
}

// ---- after an EXT build: close the holes -----------------------------------------------------------------------
// info[0] = T (virtual positions handed out), [1] = n (entries = packed length), [2] = M (hole slots below n = entries
// at or beyond n), [3] = jn (block that holds position n), [4] = log2 of the blocks per scanning workgroup, [5] = the scan's tickets
// (zeroed by the host), [8 + g] = holes in front of workgroup g's stretch
// (several workgroups: one scanned ~400 K blocks in 0.37 ms, bound by what one CU reads and writes.  Workgroup g scans its
// stretch of blocks from zero, hpre[] holds those local sums, and the last workgroup to finish - a ticket - turns the
// stretches' totals into their bases: info[8 + g]; the holes in front of block j are ext_holes_before(j).)
constexpr uint32_t XSCAN_WGS = 32, XSCAN_CH = 1024 * 16;  // workgroups of the scan; blocks a workgroup scans per trip
__device__ __forceinline__ uint64_t ext_holes_before(const uint64_t *__restrict__ hpre, const uint64_t *__restrict__ info,
                                                     uint64_t j) {
    return hpre[j] + info[8 + (j >> info[4])];  // (a stretch is a power of two of blocks: a 64-bit division per probe of ext_patch's search cost 40 %)
}
__global__ __launch_bounds__(1024) void ext_scan_kernel(ExtOut xo, uint64_t *__restrict__ hpre, uint64_t max_blocks,
                                                         uint64_t patch_grid, uint64_t *__restrict__ info,
                                                         uint32_t *__restrict__ flags) {
    __shared__ uint64_t wtot[16];
    __shared__ uint64_t carry;
    __shared__ bool last;
    const uint64_t T = *xo.extent;
    uint64_t nb = T / XBLK;
    const bool too_far = nb > max_blocks || (T > xo.max && T - xo.max > xo.ovf_cap);
    if (too_far) {  // the keys were not spread evenly enough for the scratch: the host builds the ordinary way
        if (threadIdx.x == 0 && blockIdx.x == 0) {
            atomicOr(flags, 4u);
            info[0] = info[1] = info[2] = info[3] = 0;
        }
        return;
    }
    constexpr uint32_t PT = XSCAN_CH / 1024;  // blocks per thread and trip
    const uint64_t trips = (nb + XSCAN_CH - 1) / XSCAN_CH;
    uint32_t lg = 14;  // log2 of the blocks of a workgroup's stretch: a power of two of trips
    static_assert(XSCAN_CH == 1u << 14, "a trip's blocks");
    while (((uint64_t)gridDim.x << lg) < (trips << 14)) lg++;
    const uint64_t per_wg = 1ull << lg;
    const uint64_t c_lo = (uint64_t)blockIdx.x * per_wg, c_hi = c_lo + per_wg < nb ? c_lo + per_wg : nb;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t c0 = c_lo; c0 < c_hi; c0 += XSCAN_CH) {  // hpre[j] = holes of the stretch's blocks < j
        const uint64_t i0 = c0 + (uint64_t)threadIdx.x * PT;
        uint32_t h[PT];
        uint64_t v = 0;
#pragma unroll
        for (uint32_t u = 0; u < PT; u++) {
            h[u] = i0 + u < nb ? XBLK - xo.fill[i0 + u] : 0u;
            v += h[u];
        }
        uint64_t inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            const uint64_t u = __shfl_up(inc, off, 64);
            if ((threadIdx.x & 63) >= (uint32_t)off) inc += u;
        }
        if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint64_t base = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) base += wtot[w];
        uint64_t run = base + inc - v;
#pragma unroll
        for (uint32_t u = 0; u < PT; u++) {
            if (i0 + u < nb) hpre[i0 + u] = run;
            run += h[u];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + inc;
        __syncthreads();
    }
    // the stretch's total, then a ticket: whoever draws the last one has every stretch's total in front of it
    if (threadIdx.x == 0) {
        info[8 + blockIdx.x] = carry;
        __threadfence();
        last = atomicAdd(reinterpret_cast<unsigned long long *>(&info[5]), 1ull) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) {
        __threadfence();
        uint64_t htot = 0;
        for (uint32_t g = 0; g < gridDim.x; g++) {
            const uint64_t t = __hip_atomic_load(&info[8 + g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            info[8 + g] = htot;
            htot += t;
        }
        info[8 + gridDim.x] = htot;  // (block nb, when the last stretch is full, asks one past the workgroups)
        info[4] = lg;
        const uint64_t n = nb * XBLK - htot;
        const uint64_t jn = n / XBLK;
        uint64_t M = htot;  // n == T: no hole below n that is not counted ... (jn == nb)
        if (jn < nb) {
            const uint64_t fill_end = jn * XBLK + xo.fill[jn];  // entries of block jn end here
            // (hpre[jn] was written by another workgroup: read past this CU's caches)
            const uint64_t hj = __hip_atomic_load(&hpre[jn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + info[8 + (jn >> lg)];
            M = hj + (n > fill_end ? n - fill_end : 0);
        }
        info[0] = nb * XBLK;
        info[1] = n;
        info[2] = M;
        info[3] = jn;
        if (n > xo.max) atomicOr(flags, 2u);
        // ext_patch_kernel moves the entries of blocks jn .. nb - 1, one workgroup each: more blocks than it was launched
        // with (keys spread unevenly over the build's workgroups while the caller's arrays had slack - ADVICE r3) would
        // leave entries beyond n where they are: the build is redone the ordinary way instead
        if (nb - jn > patch_grid) atomicOr(flags, 4u);
    }
}

// one workgroup per block at or beyond position n: its entries move into the hole slots below n, in order
__global__ __launch_bounds__(256) void ext_patch_kernel(ExtOut xo, const uint64_t *__restrict__ hpre,
                                                        const uint64_t *__restrict__ info,
                                                        const uint32_t *__restrict__ flags) {
    if (*flags & 6u) return;  // the arrays are too small (2) or the build is being redone (4): nothing to pack
    const uint64_t T = info[0], n = info[1], M = info[2], jn = info[3];
    const uint64_t j = jn + blockIdx.x;
    if (j * XBLK >= T) return;
    const uint64_t b0 = j * XBLK;
    const uint32_t fill = xo.fill[j];
    const uint32_t e0 = n > b0 ? (uint32_t)(n - b0) : 0u;  // block jn: only what lies at or beyond n
    const uint64_t real_before = b0 - ext_holes_before(hpre, info, j), below_n = n - M;  // entries in blocks < j; entries at positions < n
    for (uint32_t e = e0 + threadIdx.x; e < fill; e += 256) {
        const uint64_t rho = real_before + e - below_n;  // this entry's rank among the entries at or beyond n
        // the rho-th hole slot below n: the last block d <= jn with hpre[d] <= rho
        uint64_t lo = 0, hi = jn;
        while (lo < hi) {
            const uint64_t mid = (lo + hi + 1) >> 1;
            if (ext_holes_before(hpre, info, mid) <= rho) lo = mid; else hi = mid - 1;
        }
        const uint64_t dst = lo * XBLK + xo.fill[lo] + (rho - ext_holes_before(hpre, info, lo));
        uint64_t key;
        uint32_t occ;
        xo.get(b0 + e, key, occ);
        xo.keys[dst] = key;  // (dst < n <= max)
        xo.counts[dst] = occ;
    }
}

// what build could not place (a range with more distinct keys than slots): through the probing path, which reports
// the full table
__global__ __launch_bounds__(BLOCK) void spill_insert_kernel(const uint64_t *__restrict__ spill_n,
                                                             const uint64_t *__restrict__ spill_keys,
                                                             const uint32_t *__restrict__ spill_counts,
                                                             uint64_t spill_cap, TableRef t,
                                                             uint64_t *__restrict__ distinct) {
    uint64_t n = *spill_n;
    if (n > spill_cap) n = spill_cap;
    uint32_t fresh = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        const uint32_t st = kttab::table_add(t, spill_keys[i], spill_counts[i]);
        if (st == 0u) atomicOr(t.flags, 1u);
        fresh += st == 2u;
    }
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_down(fresh, o, 64);
    if ((threadIdx.x & 63) == 0 && fresh)
        atomicAdd(reinterpret_cast<unsigned long long *>(distinct), (unsigned long long)fresh);
}

// dense -> image, in place: one workgroup per range reads the range's packed entries, inserts them into the LDS image
// (distinct keys: plain CAS probing, counts carried along) and writes the whole range
__global__ __launch_bounds__(BUILD_T) void materialize_kernel(Slot *__restrict__ slots, uint32_t RS, uint64_t n_ranges,
                                                              uint32_t shift, uint32_t m8, uint32_t kbits,
                                                              const uint32_t *__restrict__ range_counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long *const skeys = reinterpret_cast<unsigned long long *>(smem_raw);
    uint32_t *const scounts = reinterpret_cast<uint32_t *>(smem_raw + (size_t)RS * 8);
    const uint32_t tid = threadIdx.x;
    for (uint64_t r = blockIdx.x; r < n_ranges; r += gridDim.x) {
        for (uint32_t i = tid; i < RS; i += BUILD_T) {
            skeys[i] = KT_EMPTY_KEY;
            scounts[i] = 0;
        }
        ktd::lds_barrier();
        const uint32_t D = range_counts[r];
        uint4 *rs = reinterpret_cast<uint4 *>(slots + r * RS);
        const uint64_t *dkeys = reinterpret_cast<const uint64_t *>(rs);  // the dense layout: keys[D], counts[D] at 8 * RS
        const uint32_t *dcounts = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(rs) + (size_t)RS * 8);
        for (uint32_t i = tid; i < D; i += BUILD_T) {
            const uint64_t key = dkeys[i];
            const uint32_t cnt = dcounts[i];
            uint32_t s = (((uint32_t)(ktd::khash_k(key, kbits) >> shift) & (S - 1)) * m8) >> 3;
            while (atomicCAS(&skeys[s], (unsigned long long)KT_EMPTY_KEY, (unsigned long long)key) != KT_EMPTY_KEY)
                s = s + 1 == RS ? 0 : s + 1;  // (D <= RS: a free slot exists)
            scounts[s] = cnt;
        }
        ktd::lds_barrier();  // (the image below overwrites what was just read: everything is in LDS by now)
        for (uint32_t i = tid; i < RS; i += BUILD_T) {
            const uint64_t key = skeys[i];
            typedef uint32_t raw4 __attribute__((ext_vector_type(4)));
            const raw4 raw = {(uint32_t)key, (uint32_t)(key >> 32), scounts[i], 0u};
            __builtin_nontemporal_store(raw, reinterpret_cast<raw4 *>(rs + i));
        }
        ktd::lds_barrier();
    }
}

// ---- dense table -> (keys, counts) ----------------------------------------------------------------------------
// Ranges are taken in tiles of 256: tile sums (coalesced), a one-workgroup scan of the tile sums, and the copy kernel
// scans the 256 counts of its tile itself - every read and write coalesced, no atomics, output in range order.
constexpr uint32_t XT = 256;
__global__ __launch_bounds__(XT) void tile_sums_kernel(const uint32_t *__restrict__ counts, uint64_t n,
                                                       uint64_t *__restrict__ tile_sums) {
    __shared__ uint32_t part[XT / 64];
    const uint64_t i = (uint64_t)blockIdx.x * XT + threadIdx.x;
    uint32_t v = i < n ? counts[i] : 0u;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = (uint64_t)part[0] + part[1] + part[2] + part[3];
}
// tile_sums[t] -> exclusive prefix; tile_sums[n_tiles] = total.  One workgroup, tiles in chunks of 1024.
__global__ __launch_bounds__(1024) void tile_scan_kernel(uint64_t *__restrict__ tile_sums, uint64_t n_tiles) {
    __shared__ uint64_t wtot[16];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t c0 = 0; c0 < n_tiles; c0 += 1024) {
        const uint64_t i = c0 + threadIdx.x;
        const uint64_t v = i < n_tiles ? tile_sums[i] : 0;
        uint64_t inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            const uint64_t u = __shfl_up(inc, off, 64);
            if ((threadIdx.x & 63) >= (uint32_t)off) inc += u;
        }
        if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint64_t base = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) base += wtot[w];
        if (i < n_tiles) tile_sums[i] = base + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sums[n_tiles] = carry;
}
__global__ __launch_bounds__(XT) void dense_export_kernel(const Slot *__restrict__ slots, uint32_t RS, uint64_t n_ranges,
                                                          const uint32_t *__restrict__ range_counts,
                                                          const uint64_t *__restrict__ tile_offsets,
                                                          uint64_t *__restrict__ out_keys,
                                                          uint32_t *__restrict__ out_counts, uint64_t max_out) {
    __shared__ uint32_t offs[XT + 1], cnt[XT], wt[XT / 64];
    const uint32_t tid = threadIdx.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * XT;
    const uint32_t c = r0 + tid < n_ranges ? range_counts[r0 + tid] : 0u;
    uint32_t inc = c;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(inc, off, 64);
        if ((tid & 63) >= (uint32_t)off) inc += u;
    }
    if ((tid & 63) == 63) wt[tid >> 6] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (tid >> 6); w++) base += wt[w];
    offs[tid] = base + inc - c;
    cnt[tid] = c;
    __syncthreads();
    const uint64_t t0 = tile_offsets[blockIdx.x];
    // every wave copies ranges of its own (no workgroup barrier in the loop), four loads in flight per lane
    for (uint32_t r = tid >> 6; r < XT && r0 + r < n_ranges; r += XT / 64) {
        const uint32_t D = cnt[r];
        const uint64_t o = t0 + offs[r];
        const char *rs = reinterpret_cast<const char *>(slots + (r0 + r) * RS);
        const uint64_t *dkeys = reinterpret_cast<const uint64_t *>(rs);
        const uint32_t *dcounts = reinterpret_cast<const uint32_t *>(rs + (size_t)RS * 8);
        for (uint32_t i0 = tid & 63u; i0 < D; i0 += 256) {
            uint64_t vk[4];
            uint32_t vc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + 64u * u;
                vk[u] = i < D ? dkeys[i] : 0;
                vc[u] = i < D ? dcounts[i] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + 64u * u;
                if (i < D && o + i < max_out) {
                    out_keys[o + i] = vk[u];
                    out_counts[o + i] = vc[u] + 1u;  // stored value is occurrences - 1
                }
            }
        }
    }
}

uint64_t env_u64(const char *name, uint64_t dflt) {
    const char *s = getenv(name);
    if (!s || !*s) return dflt;
    return strtoull(s, nullptr, 10);
}

}  // namespace

// ---- host side: a job = plan + buffers + level 1 over one or more sources + level 2 + range build ----------------
enum SourceKind { SRC_READS, SRC_KEYS, SRC_RECORDS, SRC_PACKED };
struct SourceRec {  // what level 1 ran over (kept so that a skewed batch can be redone with exact offsets)
    SourceKind kind;
    ReadsSource rs;
    KeysSource ks;
    RecordSource rc;
    PackedSource ps;
    uint64_t n_units;  // upper bound (device-side counts may make it smaller)
};
// f(the source of r, as its own type)
template <class F>
static int with_source(const SourceRec &r, F &&f) {
    switch (r.kind) {
        case SRC_READS: return f(r.rs);
        case SRC_KEYS: return f(r.ks);
        case SRC_PACKED: return f(r.ps);
        default: return f(r.rc);
    }
}

struct BulkKnobs {  // the KT_BULK_* / KT_S1_* / KT_P2_* / KT_BUILD_* environment, read once per job (kt_bulk_begin)
    uint64_t bulk, min_bases, merge_div, g_mult, paged, fixed2, p2_big64, p2_big32, build_wgs, dense,
        verbose, ext_ovf_blocks, build_wgs_ext, direct, pack, p2_fast, p2_grid, p2_swwc, b1, build_lists, s1y;
};
static BulkKnobs read_knobs() {
    BulkKnobs k;
    k.bulk = env_u64("KT_BULK", 1);
    k.min_bases = env_u64("KT_BULK_MIN_BASES", 4ull << 20);
    k.merge_div = env_u64("KT_BULK_MERGE_DIV", 8);
    k.g_mult = env_u64("KT_BULK_G_MULT", 1);
    k.paged = env_u64("KT_BULK_PAGED", 1);
    k.fixed2 = env_u64("KT_BULK_FIXED2", 1);
    k.s1y = env_u64("KT_S1Y", 1);  // level 1 with two sort buffers and the copy-out spread over the next round (0: scatter1x)
    k.p2_big64 = env_u64("KT_P2_BIG64", 1);
    k.p2_big32 = env_u64("KT_P2_BIG32", 0);
    k.p2_grid = env_u64("KT_P2_GRID", 0);  // workgroups of the level-2 launch (0: one per bucket)
    k.b1 = env_u64("KT_BULK_B1", 0);
    k.p2_swwc = env_u64("KT_P2_SWWC", 1);  // level 2 with one line per fine bucket in LDS, whole lines written (0: sort buffer)
    k.p2_fast = env_u64("KT_P2_FAST", 1);  // 0: the general level-2 kernel also for fixed fine regions (A/B, tests)
    k.build_wgs = env_u64("KT_BUILD_WGS", 64);
    k.build_wgs_ext = env_u64("KT_BUILD_WGS_EXT", 16);
    k.build_lists = env_u64("KT_BUILD_LISTS", 1);  // 0: the dense build packs the image instead of keeping claim lists (A/B)
    k.dense = env_u64("KT_BULK_DENSE", 1);
    k.direct = env_u64("KT_BUILD_DIRECT", 1);
    k.pack = env_u64("KT_BULK_PACK", 1);  // 0: level 1 stages the reads itself (ReadsSource) instead of reading packed items  // 0: tables of exactly 4^k slots are built by probing too (A/B, tests)
    k.verbose = env_u64("KT_BULK_VERBOSE", 0);
    k.ext_ovf_blocks = env_u64("KT_EXT_OVF_BLOCKS", 0);  // tests: n + 1 = blocks of scratch behind the export target
    return k;
}

struct kt_bulk_job {
    Plan p{};
    BulkKnobs kn{};
    Meta m{};
    bool narrow = false;  // 32-bit keys through the partition passes (k <= 16)
    bool paged = false, merge = false, open = false;
    bool xcd = false;  // paged level 1 appends through per-XCD cursors (scatter1x_kernel + xcd_tails_kernel)
    uint64_t max_keys = 0, added_bound = 0;
    std::vector<SourceRec> srcs;
    kt_seg_src p2_src{};  // where part2 reads every bucket from: the level-1 output (host copy of its one device entry)
    // record sources (kt_bulk_add_records): their descriptors on the device (ctr->b_desc), the host copies the uploads read
    size_t desc_used = 0;
    std::vector<std::vector<uint64_t>> desc_host;
    size_t ksz() const { return narrow ? 4 : 8; }
};

void kt_bulk_job_free(kt_bulk_job *job) { delete job; }

namespace {

template <class K>
int level1_paged(kt_ctr *ctr, kt_bulk_job &j, const SourceRec &r) {
    kt_ctx *ctx = ctr->ctx;
    K *keys1 = (K *)ctr->b_keys1.p;
    if (!j.xcd) return kt::fail(KT_ERR_ARG, "bulk build: paged level 1 without its cursors");
    unsigned long long *xcur = j.m.xcur;
    const Plan p = j.p;
    uint32_t *const ovf = j.m.ovf;
    unsigned long long *const dump = j.m.dump;
    // two sort buffers, the copy-out spread over the next round - or, a shape that does not fit the LDS, one buffer
    const bool two = j.kn.s1y && Scatter1YShared<K, 1024>::bytes(j.p.B1) <= 160 * 1024;
    const uint32_t mult = (uint32_t)(j.kn.g_mult ? j.kn.g_mult : 1);
    const uint32_t cus = (uint32_t)ctx->n_cu > 2 * ctr->l1_spare_cus ? (uint32_t)ctx->n_cu - ctr->l1_spare_cus : (uint32_t)ctx->n_cu;
    return with_source(r, [&](const auto &src) -> int {
        using S = std::decay_t<decltype(src)>;
        if (two) {
            constexpr int T = 1024;
            const size_t lds = Scatter1YShared<K, T>::bytes(p.B1);
            auto kern = scatter1y_kernel<S, K, T>;
            KT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(cus * mult), dim3(T), lds, ctx->stream, src, p, xcur, ovf, keys1, dump);
        } else {
            constexpr int T = KT_S1X_T;
            const size_t lds = sizeof(Scatter1XShared<K, T>);
            auto kern = scatter1x_kernel<S, K, T>;
            KT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(cus * (1024 / T) * mult), dim3(T), lds, ctx->stream, src, p, xcur, ovf, keys1, dump);
        }
        KT_HIP(hipGetLastError());
        return KT_OK;
    });
}

template <class K>
int level1_exact(kt_ctr *ctr, kt_bulk_job &j) {  // hist1 over every source, scan1, scatter1 over every source
    kt_ctx *ctx = ctr->ctx;
    K *keys1 = (K *)ctr->b_keys1.p;
    const Plan p = j.p;
    const Meta m = j.m;
    KT_HIP(hipMemsetAsync(m.H, 0, (size_t)p.G * p.B1 * 4, ctx->stream));
    for (const SourceRec &r : j.srcs) {
        const int rc = with_source(r, [&](const auto &src) -> int {
            using S = std::decay_t<decltype(src)>;
            hipLaunchKernelGGL(hist1_kernel<S>, dim3(p.G), dim3(BLOCK), 0, ctx->stream, src, p, m.H);
            return KT_OK;
        });
        if (rc) return rc;
    }
    hipLaunchKernelGGL(scan1_kernel, dim3(1), dim3(1024), 0, ctx->stream, m.H, p, m.O, m.bstart);
    for (const SourceRec &r : j.srcs) {
        const int rc = with_source(r, [&](const auto &src) -> int {
            using S = std::decay_t<decltype(src)>;
            auto kern = scatter1_kernel<S, K>;
            KT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(Scatter1Shared<K>)));
            hipLaunchKernelGGL(kern, dim3(p.G), dim3(BLOCK), sizeof(Scatter1Shared<K>), ctx->stream, src, p, m.O, keys1);
            return KT_OK;
        });
        if (rc) return rc;
    }
    KT_HIP(hipGetLastError());
    return KT_OK;
}

template <class K>
int finish_typed(kt_ctr *ctr, kt_bulk_job &j) {
    kt_ctx *ctx = ctr->ctx;
    Plan &p = j.p;
    Meta &m = j.m;
    K *keys1 = (K *)ctr->b_keys1.p, *keys2 = (K *)ctr->b_keys2.p;
    if (j.paged) {
        hipLaunchKernelGGL(xcd_tails_kernel<K>, dim3(p.B1), dim3(BLOCK), 0, ctx->stream, p, (const unsigned long long *)m.xcur, m.gcur, keys1);
        // the one host round trip of the build: did every bucket fit its region?
        uint32_t ovf = 0;
        KT_HIP(hipMemcpyAsync(&ovf, m.ovf, 4, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipStreamSynchronize(ctx->stream));
        if (ovf) {  // skewed batch: exact offsets after all, and no more paged attempts on this table
            j.paged = false;
            ctr->paged_failed = true;
            p.cap1 = 0;
            p.cap2 = 0;
            if (j.added_bound > ctr->b_keys1.cap / sizeof(K)) {  // (the paged room is always the larger)
                return kt::fail(KT_ERR_NOMEM, "bulk build: key buffers too small for the exact layout");
            }
            if (int rc = level1_exact<K>(ctr, j)) return rc;
        }
    }
    // where part2 finds every bucket: the level-1 output
    P2In in{keys1, m.bstart, nullptr, 0};
    if (j.paged) {
        j.p2_src = kt_seg_src{keys1, m.gcur, p.cap1};
        KT_HIP(hipMemcpyAsync(m.srcs, &j.p2_src, sizeof(kt_seg_src), hipMemcpyHostToDevice, ctx->stream));
        in.srcs = m.srcs;
        in.n_src = 1;
    }
    const uint32_t nd = p.B1;
    // one level-2 launch: `pp` says which hash bits it sorts by and how many buckets it reads, `src` where from
    // (`fail`: one word per bucket of the launch - zeroed here)
    auto run_part2 = [&](const Plan &pp, const P2In &src, K *out, uint64_t *fs, uint64_t *fe, uint32_t *fail) -> int {
        // (one line per fine bucket wants about a line's worth of keys per bucket and chunk: fan-outs below 512 keep the sort buffer)
        if (pp.cap2 && src.srcs && j.kn.p2_swwc && pp.B2 >= 512 && SwwcShared<K>::bytes(pp.B2) <= 160 * 1024) {
            // fixed fine regions, whole lines only (part2_swwc_kernel); then the general kernel over the buckets that did not
            // fit their regions (none, normally)
            const size_t lds = SwwcShared<K>::bytes(pp.B2);
            KT_HIP(hipMemsetAsync(fail, 0, ((size_t)pp.B1 + 1) * 4, ctx->stream));
            auto swwc = part2_swwc_kernel<K>;
            KT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(swwc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            uint32_t grid = pp.B1;
            if (j.kn.p2_grid && grid > j.kn.p2_grid) grid = (uint32_t)j.kn.p2_grid;
            hipLaunchKernelGGL(swwc, dim3(grid), dim3(swwc_t<K>()), lds, ctx->stream, src, pp, out, fs, fe, fail);
            KT_HIP(hipGetLastError());
            const bool bigr = (sizeof(K) == 8 ? j.kn.p2_big64 : j.kn.p2_big32) != 0 && Part2Shared<K, true>::bytes(pp.B2) <= 160 * 1024;
            auto redo_launch = [&](auto big) -> int {
                constexpr bool BIG = decltype(big)::value;
                const size_t rl = Part2Shared<K, BIG>::bytes(pp.B2);
                auto redo = part2_kernel<K, false, BIG>;
                KT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(redo), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl));
                hipLaunchKernelGGL(redo, dim3(pp.B1), dim3(p2t<K, BIG>()), rl, ctx->stream, src, pp, out, fs, fe,
                                   (const uint32_t *)fail);
                KT_HIP(hipGetLastError());
                return KT_OK;
            };
            return bigr ? redo_launch(std::true_type{}) : redo_launch(std::false_type{});
        }
        const bool big2 = (sizeof(K) == 8 ? j.kn.p2_big64 : j.kn.p2_big32) != 0 &&
                          Part2Shared<K, true>::bytes(pp.B2) <= 160 * 1024;
        auto launch = [&](auto big) -> int {
            constexpr bool BIG = decltype(big)::value;
            const size_t part2_lds = Part2Shared<K, BIG>::bytes(pp.B2);
            constexpr uint32_t P2T = (uint32_t)p2t<K, BIG>();
            if (pp.cap2 && src.srcs && j.kn.p2_fast && pp.B2 <= 4 * P2T) {
                // fixed fine regions: the lean kernel, then the general one over the buckets that did not fit (none, normally)
                KT_HIP(hipMemsetAsync(fail, 0, ((size_t)pp.B1 + 1) * 4, ctx->stream));
                auto go = [&](auto pb) -> int {
                    constexpr int PB = decltype(pb)::value;
                    auto fast = part2_fast_kernel<K, BIG, PB>;
                    KT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fast), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)part2_lds));
                    uint32_t grid = pp.B1;
                    if (j.kn.p2_grid && grid > j.kn.p2_grid) grid = (uint32_t)j.kn.p2_grid;
                    hipLaunchKernelGGL(fast, dim3(grid), dim3(P2T), part2_lds, ctx->stream, src, pp, out, fs, fe, fail);
                    KT_HIP(hipGetLastError());
                    return KT_OK;
                };
                const int rc = pp.B2 <= P2T       ? go(std::integral_constant<int, 1>{})
                               : pp.B2 <= 2 * P2T ? go(std::integral_constant<int, 2>{})
                                                  : go(std::integral_constant<int, 4>{});
                if (rc) return rc;
                auto redo = part2_kernel<K, false, BIG>;
                KT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(redo), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)part2_lds));
                hipLaunchKernelGGL(redo, dim3(pp.B1), dim3(P2T), part2_lds, ctx->stream, src, pp, out, fs, fe,
                                   (const uint32_t *)fail);
                KT_HIP(hipGetLastError());
                return KT_OK;
            }
            // (16-byte loads where every region starts on a 16-byte boundary: paged level-1 outputs)
            const bool l16 = KT_P2_LOAD16 && sizeof(K) == 8 && pp.cap2 && src.srcs;
            auto part2 = !pp.cap2 ? part2_kernel<K, false, BIG> : l16 ? part2_kernel<K, true, BIG, true> : part2_kernel<K, true, BIG>;
            KT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(part2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)part2_lds));
            hipLaunchKernelGGL(part2, dim3(pp.B1), dim3(P2T), part2_lds, ctx->stream, src, pp, out, fs, fe,
                               (const uint32_t *)nullptr);
            KT_HIP(hipGetLastError());
            return KT_OK;
        };
        return big2 ? launch(std::true_type{}) : launch(std::false_type{});
    };
    if (int rc = run_part2(p, in, keys2, m.fstart, m.fend, m.fail)) return rc;
    const uint64_t n_fine = (uint64_t)nd * p.B2;
    // workgroups per CU over the launch; up to two are resident per CU.  Each takes ranges b, b + grid, ... with the next
    // one's bounds and first keys prefetched, so a few ranges per workgroup are enough - and a smaller static share evens
    // out what the compute units get (k=31: 2 / 4 / 8 / 16 / 32 / 128 per CU = 19.4 / 19.0 / 18.6 / 18.2 / 18.0 / 17.9 ms)
    const bool dense = !j.merge && j.kn.dense != 0;
    bool ext = dense && ctr->xt_keys && ctr->xt_counts;  // the packed entries go straight to the export arrays
    // (ext: the scratch behind the caller's arrays is sized by the number of workgroups - 16 per CU there: 18.2 against
    // 17.9 ms)
    uint64_t gb = (uint64_t)ctx->n_cu * (ext ? j.kn.build_wgs_ext : j.kn.build_wgs);
    if (gb > n_fine) gb = n_fine;
    size_t build_lds = (size_t)(p.m8 << (LOG2_S - 3)) * (sizeof(K) + 4);
    // the dense build's claim lists ride behind the image where that does not cost the CU its second workgroup
    p.lists = dense && j.kn.build_lists && build_lds + LIST_KEYS * 2 + 256 <= 80 * 1024 ? 1u : 0u;
    if (p.lists) build_lds += LIST_KEYS * 2;
    // (ctr k=31: dense build 29.0 ms + dense export 17.4 ms against image build 22-23.5 ms + export 27.5 ms; k=15: 13.8 +
    // 3.2 against 12.5 + 4.7 ms; profiles/r2_build_sweep.txt.  KT_BULK_DENSE=0 builds the probing image at once.)
    auto launch_build = [&](bool to_ext, const ExtOut &xo) -> int {
        auto build = j.merge ? build_kernel<K, true, false>
                     : to_ext ? build_kernel<K, false, true, true>
                     : dense  ? build_kernel<K, false, true>
                              : build_kernel<K, false, false>;
        if constexpr (sizeof(K) == 4) {
            // a table of exactly 4^k slots: the hash is a bijection onto them, the ranges' images are counters alone
            if (dense && !j.merge && j.kn.direct && p.kbits && p.n == p.kbits && p.m8 == 8)
                build = to_ext ? build_kernel<K, false, true, true, true> : build_kernel<K, false, true, false, true>;
        }
        KT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(build), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)build_lds));
        hipLaunchKernelGGL(build, dim3((uint32_t)gb), dim3(BUILD_T), build_lds, ctx->stream, (const K *)keys2, m.fstart,
                           m.fend, p, (Slot *)ctr->slots, m.spill_n, m.spill_keys, m.spill_counts, m.spill_cap, ctr->flags,
                           ctr->distinct, ctr->range_counts, xo);
        KT_HIP(hipGetLastError());
        return KT_OK;
    };
    if (ext) {
        // scratch behind the caller's arrays (blocks + holes reach past the packed length), the blocks' fill, the holes' prefix.
        // Holes: about half a block per workgroup; the rest of the scratch is the tolerance for workgroups that need more
        // blocks than the average (KT_EXT_OVF_BLOCKS: tests make it too small on purpose).
        const uint64_t ovf_cap = (j.kn.ext_ovf_blocks ? j.kn.ext_ovf_blocks - 1 : 2 * gb + 2) * XBLK;
        const uint64_t max_blocks = (ctr->xt_max + ovf_cap) / XBLK + 2;
        size_t off = 0;
        const size_t off_fill = off;  off += (max_blocks * 4 + 255) & ~(size_t)255;
        const size_t off_hpre = off;  off += ((max_blocks + 1) * 8 + 255) & ~(size_t)255;
        const size_t off_info = off;  off += ((8 + XSCAN_WGS + 1) * 8 + 255) & ~(size_t)255;
        const size_t off_ok = off;    off += (ovf_cap * 8 + 255) & ~(size_t)255;
        const size_t off_oc = off;    off += (ovf_cap * 4 + 255) & ~(size_t)255;
        if (int rc = ctr->b_ext.reserve(off)) return rc;
        char *xb = (char *)ctr->b_ext.p;
        uint64_t *hpre = (uint64_t *)(xb + off_hpre), *xinfo = (uint64_t *)(xb + off_info);
        const ExtOut xo{ctr->xt_keys, ctr->xt_counts, ctr->xt_max, (uint64_t *)(xb + off_ok), (uint32_t *)(xb + off_oc),
                        ovf_cap,      ctr->cursor,    (uint32_t *)(xb + off_fill), max_blocks};
        KT_HIP(hipMemsetAsync(ctr->cursor, 0, 8, ctx->stream));
        KT_HIP(hipMemsetAsync(xo.fill, 0, max_blocks * 4, ctx->stream));
        if (int rc = launch_build(true, xo)) return rc;
        // close the holes: the entries beyond the packed length move into them
        const uint64_t patch_grid = ovf_cap / XBLK + 3;
        KT_HIP(hipMemsetAsync(xinfo, 0, (8 + XSCAN_WGS + 1) * 8, ctx->stream));  // (the scan's tickets)
        hipLaunchKernelGGL(ext_scan_kernel, dim3(XSCAN_WGS), dim3(1024), 0, ctx->stream, xo, hpre, max_blocks, patch_grid, xinfo, ctr->flags);
        hipLaunchKernelGGL(ext_patch_kernel, dim3((uint32_t)patch_grid), dim3(256), 0, ctx->stream, xo,
                           (const uint64_t *)hpre, (const uint64_t *)xinfo, (const uint32_t *)ctr->flags);
        KT_HIP(hipGetLastError());
        // did the blocks fit (flag 4: keys spread too unevenly over the workgroups)?  One 4-byte read; the caller's
        // next step - kt_ctr_size, kt_ctr_export - waits for these kernels anyway.
        uint32_t fl = 0, l2_redone = 0;
        KT_HIP(hipMemcpyAsync(&fl, ctr->flags, 4, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipMemcpyAsync(&l2_redone, m.fail + p.B1, 4, hipMemcpyDeviceToHost, ctx->stream));  // (rides on the same round trip)
        KT_HIP(hipStreamSynchronize(ctx->stream));
        ctr->l2_redone = l2_redone;
        ctr->l2_buckets = p.B1;
        if (fl & 6u) {
            // 4: the blocks did not fit (or could not all be patched): the ranges are built the ordinary way and
            //    kt_ctr_export copies them out.
            // 2: the caller's arrays are smaller than the table: the same, so that the table is NOT lost - the counts
            //    stay in the table's own slots, kt_ctr_size / kt_ctr_export into large enough arrays work - and the
            //    call that counted reports KT_ERR_ARG once the build is done.
            if (fl & 2u) ctr->xt_too_small = true;
            fl &= ~6u;
            KT_HIP(hipMemcpyAsync(ctr->flags, &fl, 4, hipMemcpyHostToDevice, ctx->stream));
            KT_HIP(hipMemsetAsync(ctr->distinct, 0, 8, ctx->stream));  // (a dense build starts from an empty table)
            KT_HIP(hipStreamSynchronize(ctx->stream));                 // (fl lives on this stack frame)
            ext = false;
            if (j.kn.verbose) fprintf(stderr, "[bulk] export-target build redone into the table (%s)\n",
                                      ctr->xt_too_small ? "arrays too small" : "uneven blocks");
        }
    }
    if (!ext)
        if (int rc = launch_build(false, ExtOut{})) return rc;
    ctr->dense = dense;
    ctr->dense_ext = ext;
    if (!dense) {
        TableRef t{(Slot *)ctr->slots, ktl::geom_of(ctr), ctr->flags};
        hipLaunchKernelGGL(spill_insert_kernel, dim3(ctx->n_cu), dim3(BLOCK), 0, ctx->stream, m.spill_n, m.spill_keys,
                           m.spill_counts, m.spill_cap, t, ctr->distinct);
    }
    KT_HIP(hipGetLastError());
    if (j.kn.verbose) {
        uint64_t spilled = 0;
        KT_HIP(hipMemcpyAsync(&spilled, m.spill_n, 8, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "[bulk] k=%d keys<=%llu level1=%s level2=%s %s spilled=%llu\n", ctr->k,
                (unsigned long long)j.max_keys, j.paged ? "paged" : "exact", p.cap2 ? "fixed" : "exact",
                j.merge ? "merge" : "build", (unsigned long long)spilled);
    }
    return KT_OK;
}

}  // namespace

// The cursor sets of scatter1x: one per XCD that runs workgroups of this context's device.  A census launch (once per
// context) notes which XCC_IDs there are; id -> set is their rank, the number of sets the next power of two (a device
// in a partitioned mode may show one XCD: one set then, and a region is one plain stream).
static int xcc_sets(kt_ctx *ctx, uint32_t *nxs, uint32_t *xmap) {
    if (!ctx->xcc_n) {
        if (int rc = ctx->s_aux2.reserve(256)) return rc;
        uint32_t *mask = (uint32_t *)ctx->s_aux2.p, h = 0;
        KT_HIP(hipMemsetAsync(mask, 0, 4, ctx->stream));
        hipLaunchKernelGGL(xcc_census_kernel, dim3((uint32_t)ctx->n_cu * 4), dim3(64), 0, ctx->stream, mask);
        KT_HIP(hipGetLastError());
        KT_HIP(hipMemcpyAsync(&h, mask, 4, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipStreamSynchronize(ctx->stream));
        if (!(h & 0xFFu)) h = 1;
        uint32_t n = 0, map = 0;
        for (uint32_t id = 0; id < 8; id++)
            if (h >> id & 1u) map |= (n++) << (4 * id);
        ctx->xcc_n = n;
        ctx->xcc_map = map;
    }
    uint32_t b = 0;
    while ((1u << b) < ctx->xcc_n) b++;
    if (const char *e = getenv("KT_S1X_SETS")) {  // (experiments / tests: fewer sets than XCDs - ids fold onto them)
        const uint32_t want = (uint32_t)atoi(e);
        uint32_t wb = 0;
        while ((1u << wb) < want && wb < 3) wb++;
        if (wb < b) {
            uint32_t map = 0;
            for (uint32_t id = 0; id < 8; id++) map |= (((ctx->xcc_map >> (4 * id)) & 7u) & ((1u << wb) - 1u)) << (4 * id);
            *nxs = wb;
            *xmap = map;
            return KT_OK;
        }
    }
    *nxs = b;
    *xmap = ctx->xcc_map;
    return KT_OK;
}

// Plans the partition of at most `max_keys` k-mers into the table's ranges and reserves the buffers.  *eligible = 0:
// the table shape or the batch does not suit the bulk path (or HBM is short) and the caller uses the probing path.
// keys of room of a level-1 region for at most max_keys k-mers in B1 buckets: the bucket's share + 1/8 (the hash's own spread
// and the cursor sets' uneven fills) + 16 lines per set, a multiple of 256
static uint64_t region_room(uint64_t max_keys, uint32_t B1) {
    return (max_keys / B1 + max_keys / B1 / 8 + 16 * 256 + 255) / 256 * 256;
}

static int plan_job(kt_ctr *ctr, uint64_t max_keys, int *eligible) {
    *eligible = 0;
    kt_ctx *ctx = ctr->ctx;
    const BulkKnobs kn = read_knobs();
    if (kn.bulk == 0) return KT_OK;
    if (max_keys < kn.min_bases) return KT_OK;  // small batches: atomics are fine
    if (!ctr->job) ctr->job = new (std::nothrow) kt_bulk_job();
    if (!ctr->job) return kt::fail(KT_ERR_NOMEM, "bulk build: host alloc");
    kt_bulk_job &j = *ctr->job;
    ctr->stage_n = 0;  // (a job changes the table: kt_ctr_export_stage's entries are stale)
    j.kn = kn;
    j.open = false;
    j.srcs.clear();
    j.merge = !ctr->empty;  // the table holds data: every range is rebuilt from what it has + the batch
    if (j.merge)
        if (int rc = kt_table_image(ctr)) return rc;  // (a densely packed table gets its probing image first)
    j.narrow = ctr->k <= 16 && ctr->kbits != 0;  // (kt_ctr_create: every table of k <= 16 hashes with ktd::nhash)
    j.max_keys = max_keys;
    j.added_bound = 0;
    const size_t ksz = j.narrow ? 4 : 8;
    Plan p{};
    p.n = 64 - ctr->shift;
    p.kbits = ctr->kbits;
    p.m8 = ctr->m8;
#if KT_ABLATION
    p.dbg = (uint32_t)env_u64("KT_BUILD_DBG", 0);  // bits 0-7: build_kernel, 8-11: part2, 12-15: scatter1x
#endif
    if (p.n < LOG2_S + 2 || p.n > LOG2_S + 21) return KT_OK;
    // a rebuild moves the whole table: small batches are cheaper through the atomics
    if (j.merge && max_keys < ctr->cap / (kn.merge_div ? kn.merge_div : 1)) return KT_OK;
    const uint32_t fb = p.n - LOG2_S;
    p.b1 = (fb + 1) / 2;
    if (p.b1 > 10) p.b1 = 10;  // level 1 keeps its per-digit LDS arrays at 1024 entries
    // level 2's write-combining kernel wants a fan-out of 512 or 1024 (one line per fine bucket; below that the sort-buffer
    // kernel runs): tables of 2^16 .. 2^17 ranges give level 2 nine bits instead of eight (ctr k=15, 2^17 ranges: level 1 with
    // 256 instead of 512 buckets 23.7 -> 23.3 ms, level 2 14.4 -> 13.8 ms)
    if (fb >= 16 && fb - p.b1 < 9) p.b1 = fb - 9;
    if (kn.b1 && kn.b1 <= 10 && kn.b1 < fb && fb - kn.b1 <= 11) p.b1 = (uint32_t)kn.b1;  // (KT_BULK_B1: the split between the levels, experiments)
    p.b2 = fb - p.b1;
    if (p.b2 > 11) return KT_OK;  // (a pass resolves at most 10 (level 1) / 11 (level 2) hash bits)
    p.B1 = 1u << p.b1;
    p.B2 = 1u << p.b2;
    const uint32_t nd = p.B1;
    // persistent level-1 workgroups (the same for every source of the job): two per CU for the per-unit kernels; the wide
    // kernel launches G / 2 of them, one resident per CU (KT_BULK_G_MULT > 1: more, shorter-lived workgroups)
    p.G = (uint32_t)ctx->n_cu * 2 * (uint32_t)(kn.g_mult ? kn.g_mult : 1);
    // paged level 1 (no hist1): room per bucket = its share of the most keys there can be + 1/8 + a page per
    // workgroup (every workgroup leaves at most one partly used page per bucket)
    bool paged = kn.paged != 0 && !ctr->paged_failed;
    // (rounded to a multiple of 128 keys: every region starts on a cache line, whatever the key size)
    // (per-XCD cursors: no pages; the room beyond the keys' share covers the sets' uneven fills - every set's lines reach as
    // far as the fullest set's - and the same 1/8 for the hash's own spread; a multiple of 8 sets x 32 keys)
    const bool xcd = paged;
    const uint64_t cap1 = region_room(max_keys, p.B1), room1 = cap1;
    if (room1 * ksz >= (1ull << 31)) paged = false;  // (part2 addresses a bucket's fixed regions through 32-bit buffer offsets)
    uint64_t room_in = paged ? cap1 * p.B1 : max_keys;  // the level-1 output
    uint64_t room_out = paged ? room1 * nd : max_keys;  // keys2

    // buffers: two key arrays + metadata; if HBM is short, fall back to the incremental path
    // spill list: keys of ranges that hold more distinct keys than slots (they fail in the probing path: table full)
    const uint64_t spill_cap = 1u << 20;
    size_t meta = 0;
    const size_t off_H = meta;       meta += ((size_t)p.G * p.B1 * 4 + 255) & ~(size_t)255;
    const size_t off_O = meta;       meta += ((size_t)p.G * p.B1 * 8 + 255) & ~(size_t)255;
    const size_t off_bs = meta;      meta += ((size_t)(p.B1 + 1) * 8 + 255) & ~(size_t)255;
    const size_t off_fs = meta;      meta += (((size_t)nd * p.B2 + 1) * 8 + 255) & ~(size_t)255;
    const size_t off_fe = meta;      meta += (((size_t)nd * p.B2 + 1) * 8 + 255) & ~(size_t)255;
    const size_t off_gc = meta;      meta += ((size_t)p.B1 * 8 + 255) & ~(size_t)255;
    const size_t off_xc = meta + 0;  meta += ((size_t)8 * p.B1 * 8 + 255) & ~(size_t)255;
    const size_t off_du = meta;      meta += 1024 * 16;
    const size_t off_sr = meta;      meta += 256;
    const size_t off_ov = meta;      meta += 256;
    const size_t off_fl = meta;      meta += (((size_t)nd + 2) * 4 + 255) & ~(size_t)255;
    const size_t off_sn = meta;      meta += 256;
    const size_t off_sk = meta;      meta += (spill_cap * 8 + 255) & ~(size_t)255;
    const size_t off_sc = meta;      meta += (spill_cap * 4 + 255) & ~(size_t)255;
    auto reserve_all = [&]() {
        return ctr->b_keys1.reserve(room_in * ksz) == KT_OK && ctr->b_keys2.reserve(room_out * ksz) == KT_OK &&
               ctr->b_meta.reserve(meta) == KT_OK;
    };
    bool have = reserve_all();
    if (!have && paged) {  // the paged layout wants 1/8 more room than the keys: try the exact one before giving up
        paged = false;
        room_in = room_out = max_keys;
        have = reserve_all();
        if (have) kt::set_error("");
    }
    if (!have) {
        ctr->b_keys1.release();
        ctr->b_keys2.release();
        ctr->b_meta.release();
        kt::set_error("");
        return KT_OK;  // not enough HBM for the bulk buffers: incremental path
    }
    char *mb = (char *)ctr->b_meta.p;
    Meta m{};
    m.H = (uint32_t *)(mb + off_H);
    m.O = (uint64_t *)(mb + off_O);
    m.bstart = (uint64_t *)(mb + off_bs);
    m.fstart = (uint64_t *)(mb + off_fs);
    m.fend = (uint64_t *)(mb + off_fe);
    m.gcur = (uint64_t *)(mb + off_gc);
    m.srcs = (kt_seg_src *)(mb + off_sr);
    m.xcur = (unsigned long long *)(mb + off_xc);
    m.dump = (unsigned long long *)(mb + off_du);
    m.ovf = (uint32_t *)(mb + off_ov);
    m.fail = (uint32_t *)(mb + off_fl);
    m.spill_n = (uint64_t *)(mb + off_sn);
    m.spill_keys = (uint64_t *)(mb + off_sk);
    m.spill_counts = (uint32_t *)(mb + off_sc);
    m.spill_cap = spill_cap;
    KT_HIP(hipMemsetAsync(m.spill_n, 0, 8, ctx->stream));
    if (paged) {
        p.cap1 = cap1;
        p.room1 = room1;
        p.cap2 = kn.fixed2 ? room1 / p.B2 / (128 / ksz) * (128 / ksz) : 0;  // (whole cache lines: every fine region starts on one)
        KT_HIP(hipMemsetAsync(m.gcur, 0, (size_t)p.B1 * 8, ctx->stream));
        KT_HIP(hipMemsetAsync(m.ovf, 0, 8, ctx->stream));
        if (xcd) {
            if (int rc = xcc_sets(ctx, &p.nxs, &p.xmap)) return rc;
            // (a batch of a few dozen units does not spread evenly over the XCDs - a handful of workgroups, some XCDs with
            // one more than others - and a set's share of a region is only an eighth of it: such batches append through
            // one set; what the sets buy, whole lines out of the L2, is nothing they would notice)
            if (max_keys < (16ull << 20) && !getenv("KT_S1X_SETS")) {
                p.nxs = 0;
                p.xmap = 0;
            }
            KT_HIP(hipMemsetAsync(m.xcur, 0, (size_t)8 * p.B1 * 8, ctx->stream));
        }
    }
    j.xcd = paged && xcd;
    j.desc_used = 0;
    j.desc_host.clear();
    j.p = p;
    j.m = m;
    j.paged = paged;
    j.open = true;
    *eligible = 1;
    return KT_OK;
}

int kt_bulk_begin(kt_ctr *ctr, uint64_t max_keys, int *eligible) { return plan_job(ctr, max_keys, eligible); }

static int job_add(kt_ctr *ctr, const SourceRec &r, uint64_t bound) {
    kt_bulk_job *job = ctr->job;
    if (!job || !job->open) return kt::fail(KT_ERR_ARG, "bulk build: no open job");
    if (job->added_bound + bound > job->max_keys) return kt::fail(KT_ERR_ARG, "bulk build: more keys than the job was planned for");
    job->added_bound += bound;
    job->srcs.push_back(r);
    if (!job->paged) return KT_OK;  // exact offsets: level 1 runs in finish, over all the sources
    return job->narrow ? level1_paged<uint32_t>(ctr, *job, r) : level1_paged<uint64_t>(ctr, *job, r);
}

// level 1 over the k-mers that start in segments [seg_lo, seg_hi) of a read batch (seg_first: ktseg::seg_index_kernel)
int kt_bulk_add_reads(kt_ctr *ctr, const uint8_t *d_bases, const uint64_t *d_offsets, const uint64_t *seg_first,
                      uint64_t n_reads, uint64_t n_seg, uint64_t seg_lo, uint64_t seg_hi, uint32_t n_parts, uint32_t part) {
    SourceRec r{};
    r.kind = SRC_READS;
    r.rs.a.bases = d_bases;
    r.rs.a.offsets = d_offsets;
    r.rs.a.seg_first = seg_first;
    r.rs.a.n_reads = n_reads;
    r.rs.a.n_seg = n_seg;
    r.rs.a.k = (uint32_t)ctr->k;
    r.rs.seg_lo = seg_lo;
    r.rs.seg_hi = seg_hi;
    r.rs.n_parts = n_parts;
    r.rs.part = part;
    r.n_units = seg_hi - seg_lo;
    return job_add(ctr, r, (seg_hi - seg_lo) * ktseg::SEG);  // at most one k-mer per base
}

// level 1 over an array of canonical k-mers (KT_EMPTY_KEY entries are skipped)
int kt_bulk_add_keys(kt_ctr *ctr, const uint64_t *d_keys, uint64_t n_keys) {
    SourceRec r{};
    r.kind = SRC_KEYS;
    r.ks = KeysSource{d_keys, n_keys};
    r.n_units = (n_keys + ktseg::SEG - 1) / ktseg::SEG;
    return job_add(ctr, r, n_keys);
}

// level 1 over records (kt_superkmer.hpp): `runs` (host) = stretches of whole blocks in device memory, in any number;
// kmers_bound: the k-mers they hold at most (0: not counted against the job's plan - the caller planned for all of them)
constexpr size_t DESC_BYTES = 1u << 18;
int kt_bulk_add_records(kt_ctr *ctr, const ktsk::RecRun *runs, uint32_t n_runs, uint64_t kmers_bound) {
    kt_bulk_job *job = ctr->job;
    if (!job || !job->open) return kt::fail(KT_ERR_ARG, "bulk build: no open job");
    if (!n_runs) return KT_OK;
    // the descriptors: n_runs x {blocks, n_rec}, then n_runs + 1 cumulative block counts
    std::vector<uint64_t> h((size_t)n_runs * 3 + 1);
    uint64_t blocks = 0;
    for (uint32_t i = 0; i < n_runs; i++) {
        h[2 * (size_t)i] = (uint64_t)(uintptr_t)runs[i].blocks;
        h[2 * (size_t)i + 1] = runs[i].n_rec;
        h[2 * (size_t)n_runs + i] = blocks;
        blocks += (runs[i].n_rec + ktsk::BLOCK_RECS - 1) / ktsk::BLOCK_RECS;
    }
    h[3 * (size_t)n_runs] = blocks;
    if (!blocks) return KT_OK;
    const size_t bytes = (h.size() * 8 + 255) & ~(size_t)255;
    if (job->desc_used + bytes > DESC_BYTES) return kt::fail(KT_ERR_ARG, "bulk build: too many record sources in one job");
    if (!ctr->b_desc.p)
        if (int rc = ctr->b_desc.reserve(DESC_BYTES)) return rc;
    char *d = (char *)ctr->b_desc.p + job->desc_used;
    job->desc_used += bytes;
    job->desc_host.push_back(std::move(h));  // (alive until the job after next begins: the upload reads it)
    const std::vector<uint64_t> &hh = job->desc_host.back();
    KT_HIP(hipMemcpyAsync(d, hh.data(), hh.size() * 8, hipMemcpyHostToDevice, ctr->ctx->stream));
    SourceRec r{};
    r.kind = SRC_RECORDS;
    r.rc.runs = (const ktsk::RecRun *)d;
    r.rc.cum = (const uint64_t *)d + 2 * (size_t)n_runs;
    r.rc.n_runs = n_runs;
    r.rc.k = (uint32_t)ctr->k;
    r.rc.n_blocks = blocks;
    r.n_units = blocks;
    return job_add(ctr, r, kmers_bound);
}

// the same records, one table_add per k-mer (kt_shard.hip, when kt_bulk_begin says the batch does not suit the passes)
int kt_ctr_count_records(kt_ctr *ctr, const ktsk::RecRun *runs, uint32_t n_runs) {
    kt_ctx *ctx = ctr->ctx;
    if (!n_runs) return KT_OK;
    std::vector<uint64_t> h((size_t)n_runs * 3 + 1);
    uint64_t blocks = 0;
    for (uint32_t i = 0; i < n_runs; i++) {
        h[2 * (size_t)i] = (uint64_t)(uintptr_t)runs[i].blocks;
        h[2 * (size_t)i + 1] = runs[i].n_rec;
        h[2 * (size_t)n_runs + i] = blocks;
        blocks += (runs[i].n_rec + ktsk::BLOCK_RECS - 1) / ktsk::BLOCK_RECS;
    }
    h[3 * (size_t)n_runs] = blocks;
    if (!blocks) return KT_OK;
    ctr->stage_n = 0;
    if (int rc = ktl::table_ready(ctr)) return rc;  // (a dense table gets its probing image, a deferred clear happens now)
    ctr->empty = false;
    if (int rc = ctx->s_aux2.reserve(h.size() * 8)) return rc;
    KT_HIP(hipMemcpyAsync(ctx->s_aux2.p, h.data(), h.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    KT_HIP(hipStreamSynchronize(ctx->stream));  // (h lives on this frame)
    RecordSource src{(const ktsk::RecRun *)ctx->s_aux2.p, (const uint64_t *)ctx->s_aux2.p + 2 * (size_t)n_runs, n_runs, (uint32_t)ctr->k, blocks};
    TableRef t{(Slot *)ctr->slots, ktl::geom_of(ctr), ctr->flags};
    hipLaunchKernelGGL(count_records_kernel, dim3(ktl::grid_for(ctx, blocks, 8)), dim3(BLOCK), 0, ctx->stream, src, t, ctr->distinct);
    KT_HIP(hipGetLastError());
    return KT_OK;
}

// level 2 + the range builds; the sources must still be readable (a skewed batch is redone from them)
int kt_bulk_finish(kt_ctr *ctr) {
    kt_bulk_job *job = ctr->job;
    if (!job || !job->open) return kt::fail(KT_ERR_ARG, "bulk build: no open job");
    job->open = false;
    if (!job->paged) {
        if (int rc = job->narrow ? level1_exact<uint32_t>(ctr, *job) : level1_exact<uint64_t>(ctr, *job)) return rc;
    }
    int rc = job->narrow ? finish_typed<uint32_t>(ctr, *job) : finish_typed<uint64_t>(ctr, *job);
    job->srcs.clear();
    if (rc == KT_OK) {
        ctr->empty = false;
        ctr->needs_clear = false;  // every slot was written
        if (ctr->xt_too_small) {   // (the table is complete and usable; the caller's export arrays hold nothing)
            ctr->xt_too_small = false;
            rc = kt::fail(KT_ERR_ARG, "kt_ctr_export_target: the arrays are smaller than the table - the counts are in the table "
                                      "(kt_ctr_size, then kt_ctr_export into arrays of that size), the target arrays hold nothing");
        }
    }
    return rc;
}

int kt_bulk_build(kt_ctr *ctr, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                  uint64_t total_bases, uint32_t n_parts, uint32_t part, int *done) {
    *done = 0;
    kt_ctx *ctx = ctr->ctx;
    int eligible = 0;
    const uint64_t n_seg = (total_bases + ktseg::SEG - 1) / ktseg::SEG;
    if (int rc = kt_bulk_begin(ctr, n_seg * ktseg::SEG, &eligible)) return rc;  // at most one k-mer per base
    if (!eligible) return KT_OK;
    // the segment index (seg_first) lives in ctx scratch; same helper kernel as the other paths
    if (int rc = ctx->s_aux0.reserve((n_seg + 2) * sizeof(uint64_t))) return rc;
    uint64_t *seg_first = (uint64_t *)ctx->s_aux0.p;
    hipLaunchKernelGGL(ktseg::seg_index_kernel, dim3((uint32_t)((n_reads + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                       d_offsets, n_reads, seg_first, n_seg);
    // A pass of its own packs the reads (16 bytes per 32 bases: codes, invalid-base mask, read-start mask - what stage_segment
    // makes of them) and level 1 reads the packed items: it stages nothing through LDS and has no barrier of its own per unit.
    // ctr k=31: pack 1.5 + level 1 11.0 ms against 13.2-13.4 with the staging inside level 1; k=15: 3.1 + 14.7 against 18.3.
    // (Not for the hash partitions of an out-of-core count - every pass would pack the same reads again - and not when the
    // half byte per base of HBM is not to be had.)
    bool packed = ctr->job->kn.pack && n_parts <= 1 && ctr->job->paged;
    if (packed && ctr->b_pack.reserve((n_seg * BLOCK + 1) * sizeof(uint4)) != KT_OK) {
        kt::set_error("");
        packed = false;
    }
    if (packed) {
        SegArgs a{d_bases, d_offsets, seg_first, n_reads, n_seg, (uint32_t)ctr->k};
        hipLaunchKernelGGL(pack_segments_kernel, dim3(ktl::grid_for(ctx, n_seg, 8)), dim3(BLOCK), 0, ctx->stream, a, (uint64_t)0, n_seg,
                           (uint4 *)ctr->b_pack.p);
        KT_HIP(hipGetLastError());
        SourceRec r{};
        r.kind = SRC_PACKED;
        r.ps = PackedSource{(const uint4 *)ctr->b_pack.p, 0, n_seg, (uint32_t)ctr->k};
        r.n_units = n_seg;
        if (int rc = job_add(ctr, r, n_seg * ktseg::SEG)) return rc;
    } else if (int rc = kt_bulk_add_reads(ctr, d_bases, d_offsets, seg_first, n_reads, n_seg, 0, n_seg, n_parts, part)) {
        return rc;
    }
    if (int rc = kt_bulk_finish(ctr)) return rc;
    *done = 1;
    return KT_OK;
}

// same construction from canonical k-mers that already are an array (KT_EMPTY_KEY entries are skipped):
// the k-mers another GPU routed here.  Counts are one per array entry.
int kt_bulk_build_keys(kt_ctr *ctr, const uint64_t *d_keys, uint64_t n_keys, int *done) {
    *done = 0;
    int eligible = 0;
    if (int rc = kt_bulk_begin(ctr, n_keys, &eligible)) return rc;
    if (!eligible) return KT_OK;
    if (int rc = kt_bulk_add_keys(ctr, d_keys, n_keys)) return rc;
    if (int rc = kt_bulk_finish(ctr)) return rc;
    *done = 1;
    return KT_OK;
}

// the probing image of a table that the last build left densely packed (in place; no-op otherwise)
int kt_table_image(kt_ctr *ctr) {
    if (!ctr->dense) return KT_OK;
    if (ctr->dense_ext) {  // the entries are the caller's export arrays: back into an empty table, as (key, count) pairs
        ctr->dense = false;
        ctr->dense_ext = false;
        return kt_ctr_reload_pairs(ctr, ctr->xt_keys, ctr->xt_counts);
    }
    kt_ctx *ctx = ctr->ctx;
    const uint32_t RS = ctr->m8 << (LOG2_S - 3);
    const uint64_t n_ranges = ctr->cap / RS;
    const size_t lds = (size_t)RS * 12;
    KT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(materialize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    uint64_t g = (uint64_t)ctx->n_cu * 4;
    if (g > n_ranges) g = n_ranges;
    hipLaunchKernelGGL(materialize_kernel, dim3((uint32_t)g), dim3(BUILD_T), lds, ctx->stream, (Slot *)ctr->slots, RS, n_ranges,
                       ctr->shift, ctr->m8, ctr->kbits, ctr->range_counts);
    KT_HIP(hipGetLastError());
    ctr->dense = false;
    return KT_OK;
}

// kt_ctr_export of a densely packed table: d_keys / d_counts are device arrays of max_out entries; *n = entries in the table
int kt_table_dense_export(kt_ctr *ctr, uint64_t *d_keys, uint32_t *d_counts, uint64_t max_out, uint64_t *n) {
    kt_ctx *ctx = ctr->ctx;
    if (ctr->dense_ext) {
        // the build wrote entries [0, *distinct) of the export target: nothing to do when that is where the caller
        // wants them, a plain copy otherwise
        KT_HIP(hipMemcpyAsync(n, ctr->distinct, 8, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipStreamSynchronize(ctx->stream));
        const uint64_t w = *n < max_out ? *n : max_out;
        if (w && d_keys != ctr->xt_keys) KT_HIP(hipMemcpyAsync(d_keys, ctr->xt_keys, w * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (w && d_counts != ctr->xt_counts) KT_HIP(hipMemcpyAsync(d_counts, ctr->xt_counts, w * 4, hipMemcpyDeviceToDevice, ctx->stream));
        KT_HIP(hipStreamSynchronize(ctx->stream));
        return KT_OK;
    }
    const uint32_t RS = ctr->m8 << (LOG2_S - 3);
    const uint64_t n_ranges = ctr->cap / RS, n_tiles = (n_ranges + XT - 1) / XT;
    if (int rc = ctx->s_aux0.reserve((n_tiles + 1) * 8)) return rc;
    uint64_t *tiles = (uint64_t *)ctx->s_aux0.p;
    hipLaunchKernelGGL(tile_sums_kernel, dim3((uint32_t)n_tiles), dim3(XT), 0, ctx->stream, ctr->range_counts, n_ranges, tiles);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, tiles, n_tiles);
    if (max_out)
        hipLaunchKernelGGL(dense_export_kernel, dim3((uint32_t)n_tiles), dim3(XT), 0, ctx->stream, (const Slot *)ctr->slots, RS,
                           n_ranges, ctr->range_counts, tiles, d_keys, d_counts, max_out);
    KT_HIP(hipGetLastError());
    KT_HIP(hipMemcpyAsync(n, tiles + n_tiles, 8, hipMemcpyDeviceToHost, ctx->stream));
    KT_HIP(hipStreamSynchronize(ctx->stream));
    return KT_OK;
}

#if KT_ABLATION
extern "C" int kt_dbg_phases(unsigned long long *out) {  // (timing builds) reads and zeroes the phase counters
    unsigned long long z[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(kt_dbg_phase), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(kt_dbg_phase), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif
