// kt_oligo.hip - per-read oligonucleotide-frequency histograms on gfx950.
//
// Replaces OligoComputer::vectorise_one (reference composition/src/oligo.rs:231-259),
// OligoCgrComputer::seq_to_kmer (composition/src/oligocgr.rs:145-163) and the python
// binding's copy (pybindings/src/oligo.rs:39-69) for a whole CSR batch of reads.
//
// Shape of the problem (k=4, 150-bp reads, f64 rows): 150 B in, 1088 B out per read - a
// streaming-store kernel; HBM-bound, no MFMA (integer histogram).  The first version was
// VALU-bound (278 VALU instructions per read) and the second serialised its compute and
// store phases (profiles/r1_oligo_*), so this one is built around two things: instruction
// count, and keeping the store stream busy while the next tile is being counted.
//
// A 256-thread workgroup owns *tiles* of R consecutive reads; their bases are one contiguous
// byte range of `bases`, treated as a flat stream.  Per tile:
//
//   B  positions: each wave takes 1008-byte chunks of the stream; a lane owns 16 consecutive
//      bases (one aligned 16-byte global load, issued one tile ahead; lane 0 is the halo for
//      lane 1).  The 16 bytes are encoded with SWAR integer ops (4 bases per instruction:
//      2-bit codes by shifts/xor, validity by a v_perm_b32 table compare), packed to 32 bits
//      of codes + 16 invalid flags, and the predecessor lane's last 8 bases arrive by one DPP
//      wave_shr:1.  Each of the 16 k-mers ending in the lane is then one v_alignbit of that
//      48-bit window: fwd(p) = sum code[p-j]*4^j, identical to the reference's rolling
//      value.  Which read a base belongs to is arithmetic for equal-length
//      tiles and a binary search of the tile's offsets otherwise, so long reads spread over
//      all lanes and waves.  Validity (invalid base in the window, read boundary inside the
//      window) is 16-bit mask arithmetic; the 16 LUT reads (rank of the canonical form, LDS)
//      and the 16 ds_add_u32 into the reads' LDS histogram rows are unconditional (they add
//      0 or 1): no branches.  (16 rather than 8 bases per lane halves the lanes of one read
//      that meet in one ds_add, i.e. the same-address conflicts, and the per-lane overhead.)
//   D  output: the R x bins counts are one contiguous block of the output matrix; each wave
//      streams its rows out as 16-byte stores (2 in flight per lane) and clears the LDS
//      rows behind.  Normalisation c / d (d = max(1, total)) uses y = RN(1/d) computed once
//      per read: q0 = c*y, r = fma(-q0, d, c), q = fma(r, y, q0).  For integers
//      c <= d < 2^32 the residual r is exact and q is the correctly rounded quotient
//      (Markstein), i.e. bit-identical to the reference's f64 division.
//
// Two barriers per tile; the next tile's offsets and bases are loaded into registers
// between B and D so HBM read latency hides under the store phase.  Structures that were
// measured and dropped (double-buffered producer/consumer wave specialisation; the per-read
// wave loop with k-1 DPP shifts) are in the git history and DESIGN.md.
#include "kt_device.hpp"
#include "kt_internal.hpp"

#include <type_traits>

namespace {

constexpr uint32_t NB = 16;          // bases per lane (one aligned 16-byte load)
constexpr uint32_t CHUNK = 63 * NB;  // new bases per wave-chunk (lane 0 is halo)
constexpr uint32_t MAX_R = 64;      // reads per tile (one lane per read in the per-read steps)

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
    uint32_t R;            // reads per tile, <= MAX_R
    uint32_t norm;
    uint32_t total_step;
    uint32_t vec_per_row;  // bins / VEC
    uint32_t vec_magic;    // ceil(2^32 / vec_per_row)
    uint32_t debug;        // KT_OLIGO_DEBUG bits: 1 skip positions, 2 skip stores (ablation only)
};

using ktd::swar4;

// number of reads r in [0, nr] with roff[r] <= T, minus one = the read containing T
// (the last one starting at or before T, which skips empty reads)
__device__ __forceinline__ uint32_t find_read(const uint64_t *roff, uint32_t nr, uint64_t T) {
    uint32_t lo = 0, hi = nr;  // invariant: roff[lo] <= T, answer in [lo, hi]
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (roff[mid] <= T) lo = mid; else hi = mid - 1;
    }
    return lo;
}

struct TileCtx {
    uint64_t r0;       // first read of the tile
    uint32_t nr;       // reads in the tile
    uint64_t off0;     // base offset of its first read
    uint64_t TL;       // number of bases of the tile
};

// What a wave knows about a tile once its offsets have arrived.
struct ProdTile {
    uint64_t r0, off0, TL;
    uint32_t nr;
    uintptr_t al0;      // NB-byte aligned address at or before the tile's first base
    uint32_t sh;        // (address of first base) - al0
    uint64_t flat_end;  // tile bases are flat bytes [sh, flat_end) from al0
    uint64_t n_chunks;
    // fast path (all reads of the tile equally long, >= NB bases, tile < 2^31 bytes): everything
    // is 32-bit and read membership is a magic division
    bool general;
    uint32_t Lr, lr_magic;
    int32_t safe_lo, safe_hi;  // flat range in which an NB-byte load stays inside the buffer
};

// lane i holds offsets[r0+i] and offsets[r0+i+1] of a tile (i < nr <= 64); issued two tiles ahead
struct OffRegs {
    uint64_t o, on;
};

__device__ __forceinline__ OffRegs load_offsets(const OligoArgs &a, uint64_t tile, uint32_t lane) {
    OffRegs r{0, 0};
    const uint64_t r0 = tile * a.R;
    if (r0 < a.n_reads) {
        const uint64_t nr = (a.n_reads - r0) < a.R ? (a.n_reads - r0) : a.R;
        if (lane < nr) {
            r.o = a.offsets[r0 + lane];
            r.on = a.offsets[r0 + lane + 1];
        }
    }
    return r;
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t l) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ ProdTile make_tile(const OligoArgs &a, uint64_t tile, const OffRegs &r, uint32_t lane,
                                              uint64_t total_bytes) {
    ProdTile t;
    t.r0 = tile * a.R;
    t.nr = (uint32_t)((a.n_reads - t.r0) < a.R ? (a.n_reads - t.r0) : a.R);
    t.off0 = readlane64(r.o, 0);
    const uint64_t off1 = readlane64(r.on, t.nr - 1);
    const uint64_t len_first = readlane64(r.on, 0) - t.off0;
    t.TL = off1 - t.off0;
    const bool differs = lane < t.nr && (r.on - r.o) != len_first;
    t.general = __ballot(differs) != 0 || len_first < NB || t.TL >= 0x7FFF0000ull;
    const uintptr_t base_addr = reinterpret_cast<uintptr_t>(a.bases);
    const uintptr_t addr0 = base_addr + t.off0;
    t.al0 = addr0 & ~(uintptr_t)(NB - 1);
    t.sh = (uint32_t)(addr0 - t.al0);
    t.flat_end = t.TL + t.sh;
    t.Lr = (uint32_t)len_first;
    t.lr_magic = 0;
    t.safe_lo = 0;
    t.safe_hi = 0;
    if (t.general) {
        t.n_chunks = (t.flat_end + CHUNK - 1) / CHUNK;
    } else {
        t.n_chunks = ((uint32_t)t.flat_end + CHUNK - 1) / CHUNK;
        t.lr_magic = (uint32_t)(0xFFFFFFFFull / t.Lr) + 1u;
        // flat coordinate q is loadable iff base <= al0 + q and al0 + q + NB <= base + total
        const int64_t lo = (int64_t)base_addr - (int64_t)t.al0;                 // <= 0 unless bases is unaligned
        const int64_t hi = (int64_t)(base_addr + total_bytes) - (int64_t)t.al0;  // may exceed int32: clamp
        t.safe_lo = (int32_t)(lo < -64 ? -64 : lo);
        t.safe_hi = (int32_t)(hi > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : hi);
    }
    return t;
}

#ifndef KT_OLIGO_ABLATION
#define KT_OLIGO_ABLATION 0  // 1 (tools/build_variant*.sh only): KT_OLIGO_DEBUG bits switch phases off at run time for
#endif                       // profiling.  The shipped library has no switch that skips work.
#if KT_OLIGO_ABLATION
#define KT_DBG(a) ((a).debug)
#else
#define KT_DBG(a) 0u
#endif

// six waves per SIMD (77 VGPRs instead of 82) and tiles small enough for six workgroups per CU: the kernel is
// a three-way balance of VALU (60 % busy), LDS (50 %) and the store stream, with waves parked half of the time
// (profiles/r1_oligo_final_pmc.txt) - one more resident workgroup measured -2.5 %, two more (spills) +8 %
#ifndef KT_OLIGO_WPE
#define KT_OLIGO_WPE 6
#endif
#define KT_OLIGO_WPE_ATTR __attribute__((amdgpu_waves_per_eu(KT_OLIGO_WPE, KT_OLIGO_WPE)))

#ifndef KT_OLIGO_GLOAD
#define KT_OLIGO_GLOAD 1
#endif
#ifndef KT_OLIGO_LBAR
#define KT_OLIGO_LBAR 1
#endif

// 16 bytes at integer address p.  The address space is spelled out: through a generic pointer this is a
// FLAT load, which counts against lgkmcnt as well as vmcnt, so every later wait for an LDS result would
// also wait for the prefetched chunks.
__device__ __forceinline__ uint4 load16(uintptr_t p) {
#if KT_OLIGO_GLOAD
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const u32x4 __attribute__((address_space(1))) *gptr;
    const u32x4 v = *(gptr)p;
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const uint4 *>(p);
#endif
}

// Workgroup barrier that orders LDS only.  __syncthreads() also drains vmcnt, i.e. it would wait for the
// rows just stored (and the chunks just prefetched) to complete before the next phase may start.
__device__ __forceinline__ void lds_barrier() {
#if KT_OLIGO_LBAR
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#else
    __syncthreads();
#endif
}

// bytes [p, p+16) with everything outside the buffer replaced by 0xFF (first / last bytes only)
__device__ __forceinline__ uint4 load_guarded(uintptr_t p, uintptr_t base_addr, uint64_t total_bytes) {
    uint32_t w[4] = {0, 0, 0, 0};
    for (int j = 0; j < (int)NB; j++) {
        const uintptr_t b = p + j;
        const uint32_t c = (b >= base_addr && b < base_addr + total_bytes)
                               ? *reinterpret_cast<const unsigned char *>(b)
                               : 0xFFu;
        w[j >> 2] |= c << (8 * (j & 3));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// lane l of chunk ci owns flat bytes [q, q+NB), q = ci*CHUNK + NB*(l-1)   (lane 0: halo)
__device__ __forceinline__ uint4 load_chunk(const OligoArgs &a, const ProdTile &t, uint64_t ci, uint32_t lane,
                                            uint64_t total_bytes) {
    const uintptr_t base_addr = reinterpret_cast<uintptr_t>(a.bases);
    uint4 v = make_uint4(0x4E4E4E4Eu, 0x4E4E4E4Eu, 0x4E4E4E4Eu, 0x4E4E4E4Eu);  // "NNNN": lanes outside the tile
#if KT_OLIGO_ABLATION
    if (KT_DBG(a) & 4u) return make_uint4(0x54474341u + lane, 0x41434754u, 0x47474343u, 0x41544154u);  // ablation
#endif
    if (!t.general) {
        const int32_t q = (int32_t)((uint32_t)ci * CHUNK) + (int32_t)NB * ((int32_t)lane - 1);
        if (q >= 0 && q < (int32_t)(uint32_t)t.flat_end) {
            const uintptr_t p = t.al0 + (uint32_t)q;
            if (q >= t.safe_lo && q + (int32_t)NB <= t.safe_hi) v = load16(p);
            else v = load_guarded(p, base_addr, total_bytes);
        }
    } else {
        const int64_t q = (int64_t)(ci * CHUNK) + (int64_t)NB * ((int64_t)lane - 1);
        if (q >= 0 && (uint64_t)q < t.flat_end) {
            const uintptr_t p = t.al0 + (uint64_t)q;
            if (p >= base_addr && p + NB <= base_addr + total_bytes) v = load16(p);
            else v = load_guarded(p, base_addr, total_bytes);
        }
    }
    return v;
}

// encode the lane's 16 bases: P = codes (base i at bits 2*(15-i)), V = invalid flags (bit 15-i)
__device__ __forceinline__ void encode16(uint4 data, uint32_t &P, uint32_t &V) {
    uint32_t pa, va, ra, pb, vb, rb, pc, vc, rc, pd, vd, rd;
    swar4(data.x, pa, va, ra);
    swar4(data.y, pb, vb, rb);
    swar4(data.z, pc, vc, rc);
    swar4(data.w, pd, vd, rd);
    P = (pa << 24) | (pb << 16) | (pc << 8) | pd;
    V = (va << 12) | (vb << 8) | (vc << 4) | vd;
    if (__ballot((ra | rb | rc | rd) != 0) != 0) {  // rare: raw 0..3 bytes present -> per-byte path
        const uint32_t w[4] = {data.x, data.y, data.z, data.w};
        P = 0;
        V = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t e = ktd::nt4((w[i >> 2] >> (8 * (i & 3))) & 0xFFu);
            P |= (e & 3u) << (2 * (15 - i));
            V |= (e >> 2) << (15 - i);
        }
    }
}

// window = the lane's 16 bases preceded by the previous lane's last 8 (one DPP): base i of this
// lane sits at bits 2*(15-i) of (Whi:P); invalid flags at bit 15-i of VV, previous lane above
__device__ __forceinline__ void halo(uint32_t P, uint32_t V, uint32_t &Whi, uint32_t &VV) {
    const uint32_t PV = (V << 16) | (P & 0xFFFFu);
    const uint32_t prevPV = ktd::wave_shr1(PV, 0xFFFF0000u);  // lane 0: "all invalid"
    Whi = prevPV & 0xFFFFu;
    VV = (prevPV & 0xFFFF0000u) | V;
}

// The lane's 16 k-mers -> histogram.  pos0 = index of base 0 in its read (< 0: before the tile),
// rem = bases left in that read from base 0 on, len1 = length of the following read (0: none);
// all clamped to small ranges by the caller.  Branch-free: 16-bit masks over the lane's bases
// (base i <-> bit 15-i), 16 unconditional LUT reads, 16 unconditional ds_add of 0 or 1.
#ifndef KT_OLIGO_LUTREG
#define KT_OLIGO_LUTREG 0  // 1 (k <= 4): the canonical-rank LUT (<= 256 one-byte bins) in ONE register per lane of the wave - lane
#endif                     // i holds the bins of k-mers 4 i .. 4 i + 3 - looked up with a ds_bpermute (crossbar, no LDS banks) +
                           // a v_perm byte select instead of a ds_read_u16 at a random LDS address.  Measured round 3, same
                           // call, twice: 2.15 / 2.18 ms against 2.08 / 2.11 ms for the LDS table - the two extra VALU
                           // instructions per k-mer cost more than the bank conflicts they remove (the kernel's B phase is
                           // issue bound).  Off.
template <int K, bool CANON>
__device__ __forceinline__ void emit16(const OligoArgs &a, uint32_t P, uint32_t V, uint32_t lane, uint32_t rid0,
                                       int32_t pos0, uint32_t rem, uint32_t len1, const uint16_t *lut, uint32_t lutreg,
                                       uint32_t *hist, uint32_t *tot) {
    constexpr uint32_t KMASK = (1u << (2 * K)) - 1u;
    const uint32_t R = a.R, bins = a.bins;
    uint32_t Whi, VV;
    halo(P, V, Whi, VV);
    // ge(x) = { i >= x } = 0xFFFF >> clamp(x, 0, 16)
    uint32_t B = VV;  // bit 15-i: some invalid base in the window ending at base i
#pragma unroll
    for (int j = 1; j < K; j++) B |= VV >> j;
    const int32_t a1 = (int32_t)(K - 1) - pos0;  // current read: bases i < a1 lack predecessors
    const uint32_t g1 = 0xFFFFu >> (uint32_t)(a1 < 0 ? 0 : (a1 > 16 ? 16 : a1));
    const uint32_t gr = 0xFFFFu >> (rem > 16u ? 16u : rem);                        // bases of the following read
    const uint32_t grk = 0xFFFFu >> (rem + (K - 1) > 16u ? 16u : rem + (K - 1));  // ... with k-1 predecessors
    const uint32_t grl = 0xFFFFu >> (rem + len1 > 16u ? 16u : rem + len1);        // past the following read
    uint32_t ok = g1 & ~(B | grl | (gr & ~grk)) & 0xFFFFu;
    ok = lane != 0 ? ok : 0u;
    const uint32_t rsafe = rid0 < R ? rid0 : R - 1;  // dead lanes add 0 to a real row
    const uint32_t has_next = rsafe + 1 < R ? 1u : 0u;
    // byte addressing throughout: the LUT already holds 4 * bin, so one k-mer costs compare + select + add3 + the
    // validity bit (it was seven VALU instructions with element indices)
    char *const hist_b = reinterpret_cast<char *>(hist);
    const uint32_t row0_b = rsafe * bins * 4u;
    const uint32_t rowstep_b = has_next ? bins * 4u : 0u;
#pragma unroll
    for (int h = 0; h < 2; h++) {  // two batches of 8: LUT reads first, then the adds
        uint32_t bin[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = h * 8 + j;
            const uint32_t f = __builtin_amdgcn_alignbit(Whi, P, 2 * (15 - i)) & KMASK;
            if constexpr (CANON && K <= 4 && KT_OLIGO_LUTREG) {
                const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(f & ~3u), (int)lutreg);  // lane f / 4's register
                bin[j] = __builtin_amdgcn_perm(0u, w, (f & 3u) | 0x0C0C0C00u) << 2;               // its byte f % 4, x 4
            } else {
                bin[j] = CANON ? (uint32_t)lut[f] : f * 4u;  // byte offset of the bin inside its row
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = h * 8 + j;
            const uint32_t step_b = ((gr >> (15 - i)) & 1u) ? rowstep_b : 0u;
            const uint32_t val = (ok >> (15 - i)) & 1u;
#if KT_OLIGO_ABLATION
            if (KT_DBG(a) & 8u) {
                if (val + bin[j] == 0xFFFFFFFFu) hist[0] = 1;  // ablation: keep the values live
                continue;
            }
#endif
            atomicAdd(reinterpret_cast<uint32_t *>(hist_b + row0_b + step_b + bin[j]), val);
        }
    }
    atomicAdd(&tot[rsafe], (uint32_t)__popc(ok & ~gr));
    atomicAdd(&tot[rsafe + has_next], (uint32_t)__popc(ok & gr));
}

// ---- one 1008-base chunk of a tile -> LDS histogram rows ------------------------------------------
template <int K, bool CANON>
__device__ __forceinline__ void process_chunk(const OligoArgs &a, const ProdTile &t, uint64_t ci, uint4 data,
                                              uint32_t lane, const uint16_t *lut, uint32_t lutreg, uint32_t *hist,
                                              uint32_t *tot, const uint64_t *roff) {
    constexpr uint32_t KMASK = (1u << (2 * K)) - 1u;
    const uint32_t bins = a.bins, nr = t.nr;
    uint32_t P, V;
    encode16(data, P, V);
    if (!t.general) {
        // equal-length tile: 32-bit positions, membership by magic division, at most one read
        // boundary inside the lane's 16 bases (len >= 16)
        const uint32_t Lr = t.Lr;
        const int32_t q = (int32_t)((uint32_t)ci * CHUNK) + (int32_t)NB * ((int32_t)lane - 1);
        const int32_t t0 = q - (int32_t)t.sh;  // tile-relative index of base 0 (< 0 in the first lanes)
        const uint32_t tt = t0 < 0 ? 0u : (uint32_t)t0;
        uint32_t rid0 = __umulhi(tt, t.lr_magic);
        if (rid0 * Lr > tt) rid0--;  // the magic quotient can overshoot by one
        const int32_t pos0 = t0 < 0 ? t0 : (int32_t)(tt - rid0 * Lr);
        const uint32_t left = Lr - (uint32_t)(pos0 < 0 ? 0 : pos0);
        const uint32_t rem = pos0 < 0 ? 255u : (left > 255u ? 255u : left);
        const uint32_t len1 = (rid0 + 1 < nr) ? (Lr > 255u ? 255u : Lr) : 0u;
        if (q < 0 || q >= (int32_t)(uint32_t)t.flat_end || rid0 >= nr) V = 0xFFFFu;
        emit16<K, CANON>(a, P, V, lane, rid0, pos0, rem, len1, lut, lutreg, hist, tot);
        return;
    }
    // general tile: 64-bit positions, membership by binary search of the tile's offsets
    const uint64_t off0 = t.off0, TL = t.TL;
    const int64_t q = (int64_t)(ci * CHUNK) + (int64_t)NB * ((int64_t)lane - 1);
    if (!(q >= 0 && (uint64_t)q < t.flat_end)) V = 0xFFFFu;
    const int64_t t0 = q - (int64_t)t.sh;
    uint32_t rid0 = 0, rem = 255u, len1 = 0;
    int32_t pos0 = 0;
    bool slow = false;
    if (t0 <= -(int64_t)NB) {  // whole lane before the tile (halo lane of chunk 0)
        pos0 = -1024;
        V = 0xFFFFu;
    } else if (t0 < 0) {
        // bases 0..(-t0-1) lie before the tile, the rest start read 0: treat read 0 as if it
        // extended backwards (negative positions never emit)
        pos0 = (int32_t)t0;
        const uint64_t e0 = roff[1];
        const uint64_t left = (e0 - off0) + (uint64_t)(-t0);
        rem = (uint32_t)(left > 255ull ? 255ull : left);
        uint64_t l1 = 0;
        if (1 < nr) l1 = roff[2] - e0;
        len1 = (uint32_t)(l1 > 255ull ? 255ull : l1);
        slow = rem < NB && 1 < nr && len1 < NB - rem;
    } else if ((uint64_t)t0 >= TL) {
        V = 0xFFFFu;
    } else {
        const uint64_t T = off0 + (uint64_t)t0;
        rid0 = find_read(roff, nr, T);
        const uint64_t s0 = roff[rid0], e0 = roff[rid0 + 1];
        const uint64_t p64 = T - s0, left = e0 - T;
        pos0 = (int32_t)(p64 > 0x7FFFFFF0ull ? 0x7FFFFFF0ull : p64);
        rem = (uint32_t)(left > 255ull ? 255ull : left);
        uint64_t l1 = 0;
        if (rid0 + 1 < nr) l1 = roff[rid0 + 2] - e0;
        len1 = (uint32_t)(l1 > 255ull ? 255ull : l1);
        // a second boundary inside these 16 bases (tiny / empty next read)
        slow = rem < NB && rid0 + 1 < nr && len1 < NB - rem;
    }
    if (__ballot(slow) == 0) {
        emit16<K, CANON>(a, P, V, lane, rid0, pos0, rem, len1, lut, lutreg, hist, tot);
        return;
    }
    // per-base path: every base finds its own read (several boundaries in 16 bases)
    uint32_t Whi, VV;
    halo(P, V, Whi, VV);
#pragma unroll 1
    for (int i = 0; i < (int)NB; i++) {
        const int64_t ti = t0 + i;
        if (lane == 0 || ti < 0 || (uint64_t)ti >= TL) continue;
        const uint64_t T = off0 + (uint64_t)ti;
        const uint32_t rid = find_read(roff, nr, T);
        const uint64_t p64 = T - roff[rid];
        const uint32_t bad = (VV >> (15 - i)) & ((1u << K) - 1u);
        if (p64 >= (uint64_t)(K - 1) && bad == 0) {
            const uint32_t f = (uint32_t)((((uint64_t)Whi << 32) | P) >> (2 * (15 - i))) & KMASK;
            const uint32_t bin_b = CANON ? (uint32_t)lut[f] : f * 4u;  // byte offset inside the row
            atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(hist) + rid * bins * 4u + bin_b), 1u);
            atomicAdd(&tot[rid], 1u);
        }
    }
}

// 16 bytes of an output row: written once and never read back by this kernel, so the store is marked
// non-temporal (-2 % on the k=4 benchmark against a plain store, same box)
template <class V>
__device__ __forceinline__ void store_row16(char *dst, const V &o) {
    static_assert(sizeof(V) == 16, "one 16-byte vector per lane");
    typedef uint32_t raw4 __attribute__((ext_vector_type(4)));
    raw4 raw;
    __builtin_memcpy(&raw, &o, 16);
    __builtin_nontemporal_store(raw, reinterpret_cast<raw4 *>(dst));
}

// ---- consumer: finished histogram rows -> output matrix ------------------------------------------
// Executed by NC consumer waves; wave `cw` streams the rows [cw*nr/NC, (cw+1)*nr/NC) out,
// clears them and their totals.  dnm/rcp slots are per read, so consumer waves never share.
template <int DT, int NC>
__device__ __forceinline__ void consume_tile(const OligoArgs &a, const TileCtx &t, uint32_t cw, uint32_t lane,
                                             uint32_t *hist, uint32_t *tot, double *dnm, double *rcp) {
    constexpr int VEC = OutVec<DT>::VEC;
    using vec_t = typename OutVec<DT>::type;
    const uint32_t nr = t.nr;
    const uint32_t row_lo = (uint32_t)(((uint64_t)cw * nr) / NC), row_hi = (uint32_t)(((uint64_t)(cw + 1) * nr) / NC);

    // per-read divisor and its reciprocal (one lane per read; rows <= MAX_R = 64)
    for (uint32_t i = row_lo + lane; i < row_hi; i += 64) {
        double d = 1.0;
        if (a.norm) {
            const double tt = (double)((uint64_t)tot[i] * a.total_step);
            d = tt > 1.0 ? tt : 1.0;  // f64::max(1, total), oligo.rs:255-257
        }
        tot[i] = 0;
        dnm[i] = d;
        rcp[i] = 1.0 / d;
    }
    // same wave wrote dnm/rcp; LDS executes a wave's accesses in order

    const uint32_t v_lo = row_lo * a.vec_per_row, v_hi = row_hi * a.vec_per_row;
    // tile-local 32-bit byte offsets from a wave-uniform base: one scalar base + a VGPR offset
    char *dstb = reinterpret_cast<char *>(reinterpret_cast<vec_t *>(a.out) + t.r0 * a.vec_per_row);
    using cnt_t = typename std::conditional<DT == KT_F64, uint2, uint4>::type;

    auto convert = [&](const cnt_t &c, double d, double y) {
        vec_t o;
        if constexpr (DT == KT_F64) {
            const double cx = (double)c.x, cy = (double)c.y;
            o.x = ktd::quot_f64(cx, d, y);
            o.y = ktd::quot_f64(cy, d, y);
        } else if constexpr (DT == KT_F32) {
            const float df = (float)d, yf = (float)y;
            const float cx = (float)c.x, cy = (float)c.y, cz = (float)c.z, cw4 = (float)c.w;
            const float qx = __fmul_rn(cx, yf), qy = __fmul_rn(cy, yf), qz = __fmul_rn(cz, yf),
                        qw = __fmul_rn(cw4, yf);
            o.x = __fmaf_rn(__fmaf_rn(-qx, df, cx), yf, qx);
            o.y = __fmaf_rn(__fmaf_rn(-qy, df, cy), yf, qy);
            o.z = __fmaf_rn(__fmaf_rn(-qz, df, cz), yf, qz);
            o.w = __fmaf_rn(__fmaf_rn(-qw, df, cw4), yf, qw);
        } else {
            o = c;
        }
        return o;
    };

    // U independent 16-byte outputs per lane per trip, unpredicated: all LDS reads first, then
    // the arithmetic, then the stores; a predicated tail finishes the wave's rows.
    constexpr int U = 2;
    uint32_t vb = v_lo;
    for (; vb + 64 * U <= v_hi; vb += 64 * U) {
        cnt_t c[U];
        double d[U], y[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t v = vb + u * 64 + lane;
            cnt_t *hp = reinterpret_cast<cnt_t *>(hist + v * VEC);
            c[u] = *hp;
            *hp = cnt_t{};
            const uint32_t r = __umulhi(v, a.vec_magic);
            d[u] = dnm[r];
            y[u] = rcp[r];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const vec_t o = convert(c[u], d[u], y[u]);
            const uint32_t v = vb + u * 64 + lane;
            if (!(KT_DBG(a) & 2u)) store_row16(dstb + (uint32_t)(v * (uint32_t)sizeof(vec_t)), o);
        }
    }
    for (uint32_t v = vb + lane; v < v_hi; v += 64) {
        cnt_t *hp = reinterpret_cast<cnt_t *>(hist + v * VEC);
        const cnt_t c = *hp;
        *hp = cnt_t{};
        const uint32_t r = __umulhi(v, a.vec_magic);
        const vec_t o = convert(c, dnm[r], rcp[r]);
        if (!(KT_DBG(a) & 2u)) store_row16(dstb + (uint32_t)(v * (uint32_t)sizeof(vec_t)), o);
    }
}

// Single-buffer variant: every wave counts its chunks of the tile (B), barrier, every wave
// streams out its rows of the same tile (D), barrier.  Half the LDS of the double-buffered
// kernel, so twice the resident workgroups; loads for the next tile are issued between the
// phases so they are in flight while the rows are being stored.
// PW ("producer wave", the big-row shapes): NW + 1 waves; wave 0 reads the input and never stores, waves 1 .. NW write the
// rows and - as long as a tile is a single chunk - never load.  Why: a wave's loads and stores complete in issue order
// on one counter, so a wave that has stored a tile's rows and then needs a loaded value waits for ALL of its stores to
// land before it may go on - with four waves per CU writing 128 KB per tile, every tile boundary emptied the CU's store
// queue (comp cgr k=7, 1 M reads: 5.84 ms; 5.24 ms with the chunk loads replaced by constants, for 0.5 % of the bytes).
// With the roles split the store waves stream from tile to tile behind nothing but the two barriers.  The producer
// publishes each tile's description (and its read offsets) in LDS a tile ahead; a tile of several chunks (long reads) has
// the store waves count chunks as well, loaded on the spot.
#ifndef KT_OLIGO_PW_AHEAD
#define KT_OLIGO_PW_AHEAD 2
#endif
template <int K, bool CANON, int DT, int NW, bool PW = false>
__device__ __forceinline__ void oligo_sb_body(const OligoArgs &a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // k <= 5: the canonical-rank LUT (<= 2 KB) lives in LDS; k >= 6 (8 / 32 KB) it is read through
    // L2 instead, because there the LDS is better spent on resident tiles (rows are 8-64 KB)
    constexpr bool LUT_LDS = CANON && K <= 5;
    constexpr uint32_t NLUT = LUT_LDS ? (1u << (2 * K)) : 0u;
    constexpr uint32_t NWT = PW ? NW + 1 : NW;  // waves of the workgroup
    constexpr uint32_t NT = NWT * 64;
    constexpr int PF = 2;  // chunk loads kept in flight per wave (R=52 x 150 bp = 8 chunks / 4 waves)

    const uint32_t R = a.R, bins = a.bins;
    const uint16_t *lut = LUT_LDS ? reinterpret_cast<const uint16_t *>(smem) : a.lut;
    uint32_t off = (NLUT * 2 + 15) & ~15u;
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + off);
    off += R * bins * 4;
    uint32_t *tot = reinterpret_cast<uint32_t *>(smem + off);
    off += MAX_R * 4;
    double *dnm = reinterpret_cast<double *>(smem + off);
    off += MAX_R * 8;
    double *rcp = reinterpret_cast<double *>(smem + off);
    off += MAX_R * 8;
    uint64_t *roff = reinterpret_cast<uint64_t *>(smem + off);
    off += (MAX_R + 1) * 8;
    ProdTile *pub = reinterpret_cast<ProdTile *>(smem + off);  // (PW) the tile the store waves meet next

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    uint32_t lutreg = 0;  // k <= 4: bins of k-mers 4 lane .. 4 lane + 3, one byte each (a.lut holds 4 * bin)
    if constexpr (CANON && K <= 4 && KT_OLIGO_LUTREG) {
#pragma unroll
        for (uint32_t jj = 0; jj < 4; jj++) {
            const uint32_t f = lane * 4 + jj;
            if (f < (1u << (2 * K))) lutreg |= ((uint32_t)a.lut[f] >> 2) << (8 * jj);
        }
    }
    for (uint32_t i = tid; i < R * bins; i += NT) hist[i] = 0;
    for (uint32_t i = tid; i < MAX_R; i += NT) tot[i] = 0;
    for (uint32_t i = tid; i < NLUT; i += NT) reinterpret_cast<uint16_t *>(smem)[i] = a.lut[i];
    __syncthreads();

    const uint64_t n_tiles = (a.n_reads + R - 1) / R;
    const uint64_t total_bytes = a.offsets[a.n_reads];
    if (blockIdx.x >= n_tiles) return;
    const uint64_t nt = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    auto tile_of = [&](uint64_t j) { return (uint64_t)blockIdx.x + j * gridDim.x; };

    if constexpr (PW) {
        constexpr int AH = KT_OLIGO_PW_AHEAD;  // tiles of input in flight ahead of the one being counted (producer wave only)
        const bool producer = wave == 0;
        OffRegs o[AH + 1];
        ProdTile tq[AH];
        uint4 cq[AH][PF];
        auto publish = [&]() {  // tq[0] / o[0] = the tile every wave meets behind the next barrier
            if (lane == 0) *pub = tq[0];
            if (lane < tq[0].nr) {
                roff[lane] = o[0].o;
                roff[lane + 1] = o[0].on;
            }
        };
        if (producer) {
#pragma unroll
            for (int i = 0; i <= AH; i++) o[i] = (uint64_t)i < nt ? load_offsets(a, tile_of(i), lane) : OffRegs{0, 0};
#pragma unroll
            for (int i = 0; i < AH; i++) {
#pragma unroll
                for (int it = 0; it < PF; it++) cq[i][it] = make_uint4(0, 0, 0, 0);
                if (i == 0 || (uint64_t)i < nt) {
                    tq[i] = make_tile(a, tile_of(i), o[i], lane, total_bytes);
#pragma unroll
                    for (int it = 0; it < PF; it++) {
                        const uint64_t ci = (uint64_t)NWT * it;
                        if (ci < tq[i].n_chunks) cq[i][it] = load_chunk(a, tq[i], ci, lane, total_bytes);
                    }
                } else {
                    tq[i] = tq[0];
                }
            }
            publish();
        }
        lds_barrier();
        for (uint64_t j = 0; j < nt; j++) {
            TileCtx tc;
            tc.r0 = tile_of(j) * R;
            tc.nr = (uint32_t)((a.n_reads - tc.r0) < R ? (a.n_reads - tc.r0) : R);
            tc.off0 = 0;
            tc.TL = 0;
            // ---- B: positions.  Chunk ci of the tile belongs to wave ci % NWT: the producer has its chunks in registers;
            // a store wave has work here only when the tile is longer than one chunk, and loads it on the spot
            if (!(KT_DBG(a) & 1u)) {
                if (producer) {
                    const ProdTile &t_cur = tq[0];
                    uint32_t it = 0;
#pragma unroll 1
                    for (uint64_t ci = 0; ci < t_cur.n_chunks; ci += NWT, it++) {
                        uint4 d;
                        if (it < (uint32_t)PF) {
                            d = cq[0][0];
#pragma unroll
                            for (int u = 1; u < PF; u++)
                                if (it == (uint32_t)u) d = cq[0][u];
                        } else {
                            d = load_chunk(a, t_cur, ci, lane, total_bytes);
                        }
                        process_chunk<K, CANON>(a, t_cur, ci, d, lane, lut, lutreg, hist, tot, roff);
                    }
                } else if ((uint64_t)wave < pub->n_chunks) {
                    const ProdTile t_cur = *pub;
#pragma unroll 1
                    for (uint64_t ci = wave; ci < t_cur.n_chunks; ci += NWT)
                        process_chunk<K, CANON>(a, t_cur, ci, load_chunk(a, t_cur, ci, lane, total_bytes), lane, lut, lutreg, hist, tot, roff);
                }
            }
            lds_barrier();
            // ---- D: the store waves write the rows out; the producer moves its queue up, requests tile j + AH and
            // publishes tile j + 1
            if (producer) {
                if (j + 1 < nt) {
#pragma unroll
                    for (int i = 0; i + 1 < AH; i++) {
                        tq[i] = tq[i + 1];
#pragma unroll
                        for (int it = 0; it < PF; it++) cq[i][it] = cq[i + 1][it];
                    }
#pragma unroll
                    for (int i = 0; i < AH; i++) o[i] = o[i + 1];
                    if (j + AH < nt) {
                        tq[AH - 1] = make_tile(a, tile_of(j + AH), o[AH - 1], lane, total_bytes);
#pragma unroll
                        for (int it = 0; it < PF; it++) {
                            const uint64_t ci = (uint64_t)NWT * it;
                            if (ci < tq[AH - 1].n_chunks) cq[AH - 1][it] = load_chunk(a, tq[AH - 1], ci, lane, total_bytes);
                        }
                    }
                    if (j + AH + 1 < nt) o[AH] = load_offsets(a, tile_of(j + AH + 1), lane);
                    publish();
                }
            } else {
                if (!(KT_DBG(a) & 16u)) __builtin_amdgcn_s_setprio(3);
                consume_tile<DT, NW>(a, tc, wave - 1, lane, hist, tot, dnm, rcp);
                __builtin_amdgcn_s_setprio(0);
            }
            lds_barrier();
        }
        return;
    }
    // How far ahead the input travels.  Small rows (k <= 5): the next tile's chunks are requested while this tile's rows
    // are stored.  Big rows (k >= 6: 4-read tiles of 16 / 32 KB rows, one 600-byte chunk per tile): TWO tiles ahead - a
    // request issued behind a saturated store queue took longer than one tile's store phase whenever the reads did not
    // happen to sit in the Infinity Cache (1 M reads of k=7: 6.05 ms against 5.5 ms with the batch resident);
    // with two tiles of slack the kernel no longer cares where its 0.5 % of input bytes come from.
#ifndef KT_OLIGO_DEEP
#define KT_OLIGO_DEEP 2  // (tiles ahead for k >= 6; 1 M reads of k=7, f32: 1 -> 6.04 ms, 2 -> 5.72, 3 -> 5.82, 4 -> 5.85)
#endif
    constexpr int AHEAD = K >= 6 ? KT_OLIGO_DEEP : 1;  // tiles of input in flight ahead of the one being counted
    // q[0] = the tile being counted, q[i] = tile j + i with its chunks in flight; o[i] = offsets of tile j + i (one further)
    OffRegs o[AHEAD + 1];
    ProdTile tq[AHEAD];
    uint4 cq[AHEAD][PF];
#pragma unroll
    for (int i = 0; i <= AHEAD; i++) o[i] = (uint64_t)i < nt ? load_offsets(a, tile_of(i), lane) : OffRegs{0, 0};
#pragma unroll
    for (int i = 0; i < AHEAD; i++) {
#pragma unroll
        for (int it = 0; it < PF; it++) cq[i][it] = make_uint4(0, 0, 0, 0);
        if (i == 0 || (uint64_t)i < nt) {
            tq[i] = make_tile(a, tile_of(i), o[i], lane, total_bytes);
#pragma unroll
            for (int it = 0; it < PF; it++) {
                const uint64_t ci = wave + (uint64_t)NW * it;
                if (ci < tq[i].n_chunks) cq[i][it] = load_chunk(a, tq[i], ci, lane, total_bytes);
            }
        } else {
            tq[i] = tq[0];
        }
    }
    for (uint64_t j = 0; j < nt; j++) {
        ProdTile &t_cur = tq[0];
        // ---- B: positions ------------------------------------------------------------------
        if (!(KT_DBG(a) & 1u)) {
            if (lane < t_cur.nr) {
                roff[lane] = o[0].o;
                roff[lane + 1] = o[0].on;
            }
            // one rolled loop = one copy of the chunk body in the binary (the unrolled form was
            // 37 KB of code and 100 VGPRs); the prefetched registers are picked by a select chain
            uint32_t it = 0;
#pragma unroll 1
            for (uint64_t ci = wave; ci < t_cur.n_chunks; ci += NW, it++) {
                uint4 d;
                if (it < (uint32_t)PF) {
                    d = cq[0][0];
#pragma unroll
                    for (int u = 1; u < PF; u++)
                        if (it == (uint32_t)u) d = cq[0][u];
                } else {
                    d = load_chunk(a, t_cur, ci, lane, total_bytes);
                }
                process_chunk<K, CANON>(a, t_cur, ci, d, lane, lut, lutreg, hist, tot, roff);
            }
        }
        TileCtx tc;
        tc.r0 = t_cur.r0;
        tc.nr = t_cur.nr;
        tc.off0 = 0;
        tc.TL = 0;
        lds_barrier();
        // ---- prefetch while this tile is stored: the queue moves up, tile j + AHEAD is requested -------------------
        if (j + 1 < nt) {
#pragma unroll
            for (int i = 0; i + 1 < AHEAD; i++) {
                tq[i] = tq[i + 1];
#pragma unroll
                for (int it = 0; it < PF; it++) cq[i][it] = cq[i + 1][it];
            }
#pragma unroll
            for (int i = 0; i < AHEAD; i++) o[i] = o[i + 1];
            if (j + AHEAD < nt) {
                tq[AHEAD - 1] = make_tile(a, tile_of(j + AHEAD), o[AHEAD - 1], lane, total_bytes);
#pragma unroll
                for (int it = 0; it < PF; it++) {
                    const uint64_t ci = wave + (uint64_t)NW * it;
                    if (ci < tq[AHEAD - 1].n_chunks) cq[AHEAD - 1][it] = load_chunk(a, tq[AHEAD - 1], ci, lane, total_bytes);
                }
            }
            if (j + AHEAD + 1 < nt) o[AHEAD] = load_offsets(a, tile_of(j + AHEAD + 1), lane);
        }
        // ---- D: rows out, histogram cleared behind ---------------------------------------------------
        // the store stream is what must never starve: waves in this phase outrank the waves of
        // other workgroups that are still counting (-5 % wall, profiles/r1_oligo_ablation.txt)
        if (!(KT_DBG(a) & 16u)) __builtin_amdgcn_s_setprio(3);
        consume_tile<DT, NW>(a, tc, wave, lane, hist, tot, dnm, rcp);
        __builtin_amdgcn_s_setprio(0);
        lds_barrier();
    }
}

// small rows (k <= 4): six waves per SIMD; larger rows keep the compiler's own register budget (the cap cost 2 %)
template <int K, bool CANON, int DT, int NW>
__global__ __launch_bounds__(NW * 64) void oligo_sb_kernel(OligoArgs a) {
    oligo_sb_body<K, CANON, DT, NW>(a);
}
template <int K, bool CANON, int DT, int NW>
__global__ __launch_bounds__(NW * 64) KT_OLIGO_WPE_ATTR void oligo_sb_kernel_dense(OligoArgs a) {
    oligo_sb_body<K, CANON, DT, NW>(a);
}

// the big-row shapes with a producer wave: NW store waves + 1
template <int K, bool CANON, int DT, int NW>
__global__ __launch_bounds__((NW + 1) * 64) void oligo_pw_kernel(OligoArgs a) {
    oligo_sb_body<K, CANON, DT, NW, true>(a);
}

using kern_t = void (*)(OligoArgs);

template <int K>
kern_t pick_pw(int count_min, int dt) {
    if (count_min) {
        switch (dt) {
            case KT_F64: return (kern_t)oligo_pw_kernel<K, true, KT_F64, 4>;
            case KT_F32: return (kern_t)oligo_pw_kernel<K, true, KT_F32, 4>;
            default: return (kern_t)oligo_pw_kernel<K, true, KT_U32, 4>;
        }
    }
    switch (dt) {
        case KT_F64: return (kern_t)oligo_pw_kernel<K, false, KT_F64, 4>;
        case KT_F32: return (kern_t)oligo_pw_kernel<K, false, KT_F32, 4>;
        default: return (kern_t)oligo_pw_kernel<K, false, KT_U32, 4>;
    }
}

template <int K, int NW>
kern_t pick_sb(int count_min, int dt) {
#define KT_PICK(CANON, DT) (K <= 4 ? (kern_t)oligo_sb_kernel_dense<K, CANON, DT, NW> : (kern_t)oligo_sb_kernel<K, CANON, DT, NW>)
    if (count_min) {
        switch (dt) {
            case KT_F64: return KT_PICK(true, KT_F64);
            case KT_F32: return KT_PICK(true, KT_F32);
            default: return KT_PICK(true, KT_U32);
        }
    }
    switch (dt) {
        case KT_F64: return KT_PICK(false, KT_F64);
        case KT_F32: return KT_PICK(false, KT_F32);
        default: return KT_PICK(false, KT_U32);
    }
#undef KT_PICK
}

template <int NW>
kern_t pick_sb_k(int k, int count_min, int dt) {
    switch (k) {
        case 3: return pick_sb<3, NW>(count_min, dt);
        case 4: return pick_sb<4, NW>(count_min, dt);
        case 5: return pick_sb<5, NW>(count_min, dt);
        case 6: return pick_sb<6, NW>(count_min, dt);
        case 7: return pick_sb<7, NW>(count_min, dt);
        default: return nullptr;
    }
}

uint32_t env_u32(const char *name, uint32_t dflt) {
    const char *s = getenv(name);
    if (!s || !*s) return dflt;
    long v = strtol(s, nullptr, 10);
    return v > 0 ? (uint32_t)v : dflt;
}

// launch tunables: read from the environment once per context (KT_KNOBS_LIVE=1 at that moment - the sweep tools -
// keeps reading them at every launch)
const kt_ctx::OligoKnobs &oligo_knobs(kt_ctx *ctx) {
    kt_ctx::OligoKnobs &kn = ctx->oligo_knobs;
    if (!kn.loaded || kn.live) {
        kn.live = env_u32("KT_KNOBS_LIVE", 0) != 0;
        kn.shape = env_u32("KT_OLIGO_SHAPE", 104);
        kn.pw = env_u32("KT_OLIGO_PW", 7);  // smallest k whose kernel runs with a producer wave (8: none)
        kn.R = env_u32("KT_OLIGO_R", 0);
        kn.oversub = env_u32("KT_OLIGO_OVERSUB", 0);
        const char *tune = getenv("KT_OLIGO_TUNE");
        kn.tune = !(tune && tune[0] == '0');
#if KT_OLIGO_ABLATION
        kn.debug = env_u32("KT_OLIGO_DEBUG", 0);
#endif
        kn.loaded = true;
    }
    return kn;
}

constexpr uint32_t TUNE_SETTINGS[kt_ctx::OligoTune::NSET] = {32, 96, 200};
constexpr int TUNE_DEFAULT = 1;

kt_ctx::OligoTune::Entry *tune_find(kt_ctx::OligoTune &tn, const void *out) {
    for (auto &e : tn.arrays)
        if (e.out == out) return &e;
    return nullptr;
}

// collects the trials whose launches have finished; an array is decided once every setting has NEED samples
void oligo_tune_poll(kt_ctx::OligoTune &tn) {
    for (auto &t : tn.ring) {
        if (!t.live || hipEventQuery(t.b) != hipSuccess) continue;
        float ms = 0.f;
        t.live = false;
        kt_ctx::OligoTune::Entry *e = tune_find(tn, t.out);
        if (!e || e->decided) continue;   // the array's entry has made room for another since
        if (hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess || !(ms > 0.f)) continue;
        e->ns_per_read[t.which] += (double)ms * 1e6 / (double)t.reads;
        e->kept[t.which]++;
    }
    (void)hipGetLastError();  // hipErrorNotReady of a pending event is not an error of ours
    for (auto &e : tn.arrays) {
        if (!e.out || e.decided) continue;
        bool all = true;
        for (int i = 0; i < tn.NSET; i++) all &= e.kept[i] >= (uint32_t)tn.NEED;
        if (all) {
            int best = TUNE_DEFAULT;
            for (int i = 0; i < tn.NSET; i++) e.ns_per_read[i] /= e.kept[i];
            for (int i = 0; i < tn.NSET; i++)   // the default unless another is clearly (1 %) faster
                if (e.ns_per_read[i] < 0.99 * e.ns_per_read[TUNE_DEFAULT] && e.ns_per_read[i] < e.ns_per_read[best]) best = i;
            e.pick = TUNE_SETTINGS[best];
            e.decided = true;
        } else if (e.launches >= (uint32_t)(tn.WARM + tn.GIVE_UP)) {  // events kept failing: the default stays
            bool pending = false;
            for (auto &t : tn.ring) pending |= t.live && t.out == e.out;
            if (!pending) {
                for (int i = 0; i < tn.NSET; i++) e.ns_per_read[i] = 0.0;
                e.decided = true;
            }
        }
    }
}

}  // namespace

// Enqueue the histogram kernel for device-resident inputs/outputs.
static int oligo_launch(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                        int k, int count_min, int norm, int total_step, int dt, void *out) {
    if (k < 3 || k > 7)  // rows beyond LDS: the global-memory path (kt_oligo_generic.hip)
        return kt_oligo_generic_launch(ctx, bases, offsets, n_reads, k, count_min, norm, total_step, dt, out);
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
    const kt_ctx::OligoKnobs &kn = oligo_knobs(ctx);
    a.debug = kn.debug;

    // reads per tile: ~22 KB of LDS histogram, at most 64 reads; the flat
    // output index v < R * vec_per_row must keep the magic division exact: v * vec_per_row < 2^32.
    // wave layout: 104 = 4 waves per workgroup, 108 = 8
    const uint32_t shape = kn.shape;
    const uint32_t nbuf = 1;
    // small rows: ~22 KB of LDS rows (6 workgroups per CU); big rows (k >= 6, 8-64 KB each) are
    // pure store streams and measured best with one large tile per CU (cfg5: R=4, 0.56 of peak)
    // small rows: what is left of a sixth of the LDS after the LUT and the per-read arrays
    const uint32_t small_budget = k <= 4 ? 26624u - 2560u - (count_min ? (2u << (2 * k)) : 0u) : 28672u;
    uint32_t R = (bins <= 1024 ? small_budget : 131072u) / (bins * 4u);
    if (R >= 8) R &= ~3u;  // k=4: 40 reads = 6 wave-chunks (1008 B) of 150-bp reads, 6 workgroups per CU
    // k=6 (8 KB rows): 6-read tiles - one wave-chunk of 150-bp reads, three workgroups per CU - instead of 12-read ones:
    // 1.70 -> 1.48 ms per 1 M reads (tools/r4_k6_sweep.py: R = 3 .. 9 -> 2.13 / 1.72 / 1.72 / 1.49 / 1.95 / 1.55 / 1.62)
    if (bins > 1024 && bins * 4u <= 8704u && R > 6) R = 6;
    if (R < 1) R = 1;
    if (R > MAX_R) R = MAX_R;
    if (kn.R) R = kn.R;
    if (R > MAX_R) R = MAX_R;
    while (R > 1 && (uint64_t)R * a.vec_per_row * a.vec_per_row >= 0x100000000ull) R--;
    a.R = R;

    size_t lds = (count_min && k <= 5) ? ((((size_t)1 << (2 * k)) * 2 + 15) & ~(size_t)15) : 0;
    lds += nbuf * (size_t)R * bins * 4;
    lds += 2 * MAX_R * 4 + 2 * MAX_R * 8 + (MAX_R + 1) * 8 + 8 + 128;  // (+ the published tile of the producer-wave kernels)
    if (lds > 160 * 1024) return kt::fail(KT_ERR_ARG, "kt_oligo_batch: tile does not fit in LDS");

    // wave specialisation: small rows are producer-heavy, big rows (k >= 6) are all output
    kern_t kern = nullptr;
    uint32_t nthreads = 256;
    switch (shape) {
        case 104: kern = pick_sb_k<4>(k, count_min, dt); nthreads = 256; break;
        case 108: kern = pick_sb_k<8>(k, count_min, dt); nthreads = 512; break;
        default: return kt::fail(KT_ERR_ARG, "kt_oligo_batch: unknown KT_OLIGO_SHAPE");
    }
    if (k >= 6 && (uint32_t)k >= kn.pw && shape == 104) {  // the big-row shapes: four store waves + a producer wave
        kern = k == 6 ? pick_pw<6>(count_min, dt) : pick_pw<7>(count_min, dt);
        nthreads = 320;
    }
    if (!kern) return kt::fail(KT_ERR_ARG, "kt_oligo_batch: k must be in 3..7");
    if (lds > 64 * 1024)
        KT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

    const uint64_t n_tiles = (n_reads + R - 1) / R;
    uint32_t per_cu = (uint32_t)((160 * 1024) / lds);
    const uint32_t by_waves = 2048 / nthreads;
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu < 1) per_cu = 1;
    // Workgroups per resident slot.  Processes on this pool come in two kinds (tools/oligo_oversub_test.py, k=4, 10 M
    // reads; same fill rate, same clocks): with 32 per slot (5 tiles per workgroup) the kernel takes 2.05 ms in one kind
    // and 2.28-2.44 ms in the other; with 96 per slot (1.7 tiles per workgroup) 2.10 and 2.06-2.21 ms, the same at 128
    // and at one tile per workgroup.  Whether a workgroup's tiles are strided by the grid or contiguous makes no
    // difference, and neither does handing tiles to long-lived workgroups through a counter (both tried): what helps
    // the slow kind is workgroups that do not live long.  k = 4 takes 96; k = 3 and k = 5 measured the same with 32 and
    // 96 in processes where k = 4 differed by 10 % (1.22 / 1.24 ms and 7.18 / 7.20 ms) and keep 32.
    // (profiles/r2_box_variance.txt)
    // The big-row shapes (k >= 6: a few reads per tile, one or two workgroups per CU) go the other way: k=7 f32, 1 M
    // reads: 2 / 4 / 8 / 16 / 32 / 64 / 128 per slot = 5.30 / 5.33 / 5.28 / 5.41 / 5.58 / 5.72 / 5.97 ms (two processes alike).
    // Since no setting is right everywhere, k = 4 measures, per output array: of the launches into it that are large
    // enough for the choice to matter, the first WARM run as before, the next ones cycle through 32 / 96 / 200 with a
    // pair of events around each, and once each has NEED samples the fastest stays for that array (96 unless another
    // is > 1 % faster; KT_OLIGO_OVERSUB fixes it, KT_OLIGO_TUNE=0 keeps 96).  The events are polled at later launches,
    // never waited for - a launch costs what it did before, and a caller that never synchronises keeps 96.
    // (Round 3, tools/r3_kind_all_vram.py: 23 outputs of 10.9 GB allocated by one process ran at 1.90 / 2.00 / 2.29 /
    // 2.40 ms with 32 per slot, 2.02 / 2.08 / 2.16 / 2.22 with 96 and 2.10 / 2.19 / 2.10 / 2.22 with 200 - four classes
    // of placement, reproducible per array, the same zero-fill rate on all of them.)
    uint32_t per_slot = kn.oversub ? kn.oversub : k == 4 ? 96 : bins > 1024 ? 8 : 32;
    kt_ctx::OligoTune &tn = ctx->oligo_tune;
    kt_ctx::OligoTune::Trial *trial = nullptr;
    const uint64_t slots = (uint64_t)ctx->n_cu * per_cu;
    // a stream that is being captured into a graph takes no event records and no event queries: such launches run with
    // what has been decided for the array so far (or the default) and take no part in the measurement
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const hipError_t cap_rc = hipStreamIsCapturing(ctx->stream, &cap);
    if (cap_rc != hipSuccess) (void)hipGetLastError();  // (the query's failure must not surface as the launch's: the launch just does not measure)
    const bool capturing = cap_rc != hipSuccess || cap != hipStreamCaptureStatusNone;
    if (k == 4 && !kn.oversub && kn.tune && (tn.paused || capturing)) {
        const kt_ctx::OligoTune::Entry *e = tune_find(tn, out);
        if (tn.forced) per_slot = tn.forced;
        else if (e && e->decided) per_slot = e->pick;
    } else if (k == 4 && !kn.oversub && kn.tune && n_tiles >= slots * TUNE_SETTINGS[TUNE_DEFAULT] && !capturing) {
        oligo_tune_poll(tn);
        kt_ctx::OligoTune::Entry *e = tune_find(tn, out);
        if (!e) {   // a new array takes the least recently used entry
            e = &tn.arrays[0];
            for (auto &x : tn.arrays)
                if (x.stamp < e->stamp) e = &x;
            *e = kt_ctx::OligoTune::Entry{};
            e->out = out;
        }
        e->stamp = ++tn.clock;
        tn.last = out;
        if (e->decided) {
            per_slot = e->pick;
        } else {
            if (e->launches >= (uint32_t)tn.WARM && e->launches < (uint32_t)(tn.WARM + tn.GIVE_UP))
                for (auto &t : tn.ring)
                    if (!t.live) { trial = &t; break; }
            if (trial && !trial->a) {  // (both events or none: a slot with one of them would be taken for a usable one)
                if (hipEventCreate(&trial->a) != hipSuccess) trial->a = nullptr;
                if (!trial->a || hipEventCreate(&trial->b) != hipSuccess) {
                    if (trial->a) (void)hipEventDestroy(trial->a);
                    trial->a = trial->b = nullptr;
                    (void)hipGetLastError();
                    trial = nullptr;
                }
            }
            if (trial) {
                const uint32_t r = (e->launches - tn.WARM) % (2 * tn.NSET);   // 0 1 2 2 1 0: a drift over the trials cancels
                trial->which = r < (uint32_t)tn.NSET ? r : 2 * tn.NSET - 1 - r;
                trial->reads = n_reads;
                trial->out = out;
                per_slot = TUNE_SETTINGS[trial->which];
            }
            e->launches++;
        }
    }
    uint64_t grid = slots * per_slot;
    if (grid > n_tiles) grid = n_tiles;
    if (grid == 0) return KT_OK;
    // the trial's events are best effort: a record that fails drops the trial, never the call
    if (trial && hipEventRecord(trial->a, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        trial = nullptr;
    }
    hipLaunchKernelGGL(kern, dim3((uint32_t)grid), dim3(nthreads), lds, ctx->stream, a);
    KT_HIP(hipGetLastError());
    if (trial) {
        if (hipEventRecord(trial->b, ctx->stream) == hipSuccess) trial->live = true;
        else (void)hipGetLastError();
    }
    return KT_OK;
}

void kt_ctx::OligoTune::release() {
    for (auto &t : ring) {
        if (t.a) (void)hipEventDestroy(t.a);
        if (t.b) (void)hipEventDestroy(t.b);
        t.a = t.b = nullptr;
        t.live = false;
    }
}

extern "C" int kt_oligo_tuning(kt_ctx *ctx, int on) {
    if (!ctx) return kt::fail(KT_ERR_ARG, "kt_oligo_tuning: null ctx");
    if (on < 0) return kt::fail(KT_ERR_ARG, "kt_oligo_tuning: mode must be 0, 1 or a number of workgroups per slot");
    ctx->oligo_tune.paused = on != 1;
    ctx->oligo_tune.forced = on >= 2 ? (uint32_t)on : 0u;
    return KT_OK;
}

extern "C" int kt_oligo_launch_info(kt_ctx *ctx, uint32_t *wgs_per_slot, int *decided, double *ns_per_read) {
    if (!ctx) return kt::fail(KT_ERR_ARG, "kt_oligo_launch_info: null ctx");
    if (int rc = ctx->use()) return rc;
    kt_ctx::OligoTune &tn = ctx->oligo_tune;
    const kt_ctx::OligoKnobs &kn = oligo_knobs(ctx);
    if (!kn.oversub && kn.tune) oligo_tune_poll(tn);
    const kt_ctx::OligoTune::Entry *e = tn.last ? tune_find(tn, tn.last) : nullptr;
    const bool done = e && e->decided && !kn.oversub && kn.tune;
    if (wgs_per_slot) *wgs_per_slot = kn.oversub ? kn.oversub : done ? e->pick : TUNE_SETTINGS[TUNE_DEFAULT];
    if (decided) *decided = done ? 1 : 0;
    if (ns_per_read)
        for (int i = 0; i < tn.NSET; i++) ns_per_read[i] = done ? e->ns_per_read[i] : 0.0;
    return KT_OK;
}

// ---- self-test of the normalisation quotient -------------------------------------------------------------
// one workgroup per divisor d (grid-stride), lanes over c = 0..d: quot_f64 against the IEEE division
__global__ __launch_bounds__(256) void quotient_check_kernel(uint32_t d_lo, uint32_t d_hi, uint64_t *__restrict__ out) {
    uint64_t bad = 0, sum = 0, n = 0;
    for (uint32_t dd = d_lo + blockIdx.x; dd <= d_hi; dd += gridDim.x) {
        const double d = (double)dd, y = 1.0 / d;   // as consume_tile computes them
        for (uint32_t c = threadIdx.x; c <= dd; c += 256) {
            const double q = ktd::quot_f64((double)c, d, y);
            const double t = __ddiv_rn((double)c, d);
            bad += __double_as_longlong(q) != __double_as_longlong(t);
            sum += (uint64_t)__double_as_longlong(t);
            n++;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        bad += __shfl_down(bad, o, 64);
        sum += __shfl_down(sum, o, 64);
        n += __shfl_down(n, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(out), (unsigned long long)n);
        atomicAdd(reinterpret_cast<unsigned long long *>(out + 1), (unsigned long long)bad);
        atomicAdd(reinterpret_cast<unsigned long long *>(out + 2), (unsigned long long)sum);
    }
}

extern "C" int kt_selftest_quotient(kt_ctx *ctx, uint32_t d_lo, uint32_t d_hi, uint64_t *n_checked,
                                    uint64_t *n_mismatch, uint64_t *checksum) {
    if (!ctx || !n_checked || !n_mismatch || !checksum) return kt::fail(KT_ERR_ARG, "kt_selftest_quotient: null");
    if (d_lo < 1 || d_hi < d_lo) return kt::fail(KT_ERR_ARG, "kt_selftest_quotient: need 1 <= d_lo <= d_hi");
    if (int rc = ctx->use()) return rc;
    if (int rc = ctx->s_aux2.reserve(64)) return rc;
    uint64_t *d_out = (uint64_t *)ctx->s_aux2.p;
    KT_HIP(hipMemsetAsync(d_out, 0, 24, ctx->stream));
    uint32_t grid = d_hi - d_lo + 1;
    if (grid > (uint32_t)ctx->n_cu * 16) grid = (uint32_t)ctx->n_cu * 16;
    hipLaunchKernelGGL(quotient_check_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_lo, d_hi, d_out);
    KT_HIP(hipGetLastError());
    uint64_t h[3] = {0, 0, 0};
    KT_HIP(hipMemcpyAsync(h, d_out, 24, hipMemcpyDeviceToHost, ctx->stream));
    KT_HIP(hipStreamSynchronize(ctx->stream));
    *n_checked = h[0];
    *n_mismatch = h[1];
    *checksum = h[2];
    return KT_OK;
}

extern "C" int kt_oligo_batch(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets,
                              uint64_t n_reads, int k, int count_min, int norm, int total_step,
                              int out_dtype, void *out, int mem) {
    if (!ctx) return kt::fail(KT_ERR_ARG, "kt_oligo_batch: null ctx");
    if (k < 1 || k > 12) return kt::fail(KT_ERR_ARG, "kt_oligo_batch: k must be in 1..12");
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
