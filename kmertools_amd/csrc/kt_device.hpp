// Device-side building blocks shared by the gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ktd {

constexpr int WAVE = 64;

// byte -> 2-bit code + validity; arithmetic form of SEQ_NT4_TABLE (reference
// kmer/src/kmer.rs:6-15): A/a 0, C/c 1, G/g 2, T/t/U/u 3, raw bytes 0..3 themselves.
// Returns code | 4 for every other byte.
__device__ __forceinline__ uint32_t nt4(uint32_t c) {
    const uint32_t idx = (c & 0xDFu) - 0x41u;              // fold case; 'A' -> 0
    // letters A(0) C(2) G(6) T(19) U(20)
    const bool letter = idx < 21u && ((0x00180045u >> idx) & 1u);
    const bool raw = c < 4u;
    const uint32_t code = raw ? c : (((c >> 1) ^ (c >> 2)) & 3u);
    return (letter || raw) ? code : (code | 4u);
}

// 4 bases in one dword -> 2-bit codes packed big-endian in 8 bits (first base in bits 7:6)
// and 4 invalid flags (first base in bit 3).  Letters ACGTU/acgtu only; raw bytes 0..3 are
// reported through `raw` (nonzero => caller takes the per-byte path).
__device__ __forceinline__ void swar4(uint32_t x, uint32_t &codes8, uint32_t &inv4, uint32_t &raw) {
    x = __builtin_bswap32(x);  // first base -> top byte, so right-shift packing is big-endian
    const uint32_t c = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
    const uint32_t y = x & 0xDFDFDFDFu;  // fold case
    // expected letter for (code, bit0): keys 4..7 -> A C G U (S0 bytes), key 3 -> T (S1 byte 3)
    const uint32_t key = c | ((x & 0x01010101u) << 2);
    const uint32_t expect = __builtin_amdgcn_perm(0x55474341u, 0x54FFFFFFu, key);
    const uint32_t diff = y ^ expect;
    const uint32_t nz = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) & 0x80808080u;  // 0x80 per bad byte
    codes8 = (c | (c >> 6) | (c >> 12) | (c >> 18)) & 0xFFu;
    inv4 = ((nz >> 7) | (nz >> 14) | (nz >> 21) | (nz >> 28)) & 0xFu;
    const uint32_t t = x & 0xFCFCFCFCu;
    raw = (t - 0x01010101u) & ~t & 0x80808080u;  // some byte < 4 (may over-report, never under)
}

// splitmix64 finaliser: table slot hash, owner hash and the synthetic-read generator
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

#ifndef KT_KHASH
#define KT_KHASH 1
#endif
// hash of a canonical k-mer: TOP bits pick the partition digits / home slot, LOW 32 bits the owning GPU.
// One 64-bit multiply (Fibonacci hashing: every product bit depends on all key bits below it, so the top
// bits are the well-mixed ones) and a fold of the high half into the low half for the owner bits.  The
// splitmix64 finaliser (two multiplies) measured 6 % slower over the whole bulk build - the hash is
// evaluated about seven times per k-mer across the partition passes - with no better bucket balance.
__host__ __device__ __forceinline__ uint64_t khash(uint64_t key) {
#if KT_KHASH == 0
    return mix64(key);
#else
    const uint64_t h = key * 0x9e3779b97f4a7c15ull;
    return h ^ (h >> 32);
#endif
}

// khash is a bijection of the 64-bit words: the key of a hash (kt_bulk.hip moves hashes through its passes)
__host__ __device__ __forceinline__ uint64_t khash_inv(uint64_t x) {
#if KT_KHASH == 0
    x = (x ^ (x >> 31) ^ (x >> 62)) * 0x319642b2d24d8ec3ull;
    x = (x ^ (x >> 27) ^ (x >> 54)) * 0x96de1b173f119089ull;
    x = x ^ (x >> 30) ^ (x >> 60);
    return x - 0x9e3779b97f4a7c15ull;
#else
    const uint64_t h = x ^ (x >> 32);
    return h * 0xf1de83e19937733dull;  // the inverse of 0x9e3779b97f4a7c15 modulo 2^64
#endif
}

// The table hash of a k-mer of k <= 16 (kbits = 2k <= 32 bits): a BIJECTION of the 2k-bit words - an odd multiply and
// a fold of the high half into the low half, inside 2k bits (the partition digits are its top bits: the multiply's, as
// with khash; the fold spreads the position inside a range) - placed in the top bits of the 64-bit hash word.  A table of
// exactly 4^k slots is then addressed by the k-mer's own hash: every canonical k-mer has a slot of its own, no key is
// stored or compared (kt_bulk.hip, the direct build; SURVEY 7 (iii): the reference's `+= 1`, counter/src/lib.rs:126-130),
// and the key comes back from the slot's index (nhash_inv).  kbits = 0: khash.  (Two rounds measured +2.2 ms on ctr k=15's
// partition passes - the hash is evaluated six times per k-mer there - for no better balance.)
// (held top-aligned in a 32-bit word - nhash_top: the product shifted up by 32 - 2k, which is also the mask; what the fold
// drags below the 2k valid bits is never looked at - so that the hash word of the partition passes is four 32-bit
// instructions and its digits one more shift each)
__host__ __device__ __forceinline__ uint32_t nhash_top(uint32_t x, uint32_t kbits) {
    const uint32_t t = (x * 0x9E3779B1u) << (32u - kbits);
    return t ^ (t >> (kbits >> 1));
}
__host__ __device__ __forceinline__ uint32_t nhash(uint32_t x, uint32_t kbits) { return nhash_top(x, kbits) >> (32u - kbits); }
__host__ __device__ __forceinline__ uint32_t nhash_inv(uint32_t y, uint32_t kbits) {
    uint32_t t = y << (32u - kbits);
    t ^= t >> (kbits >> 1);             // (the fold is its own inverse: twice the shift is the whole word)
    return ((t >> (32u - kbits)) * 0x0E8B2F51u) & (kbits >= 32u ? 0xFFFFFFFFu : (1u << kbits) - 1u);  // 0x9E3779B1^-1 mod 2^32
}
__host__ __device__ __forceinline__ uint64_t khash_k(uint64_t key, uint32_t kbits) {
    return kbits ? (uint64_t)nhash_top((uint32_t)key, kbits) << 32 : khash(key);
}

// owner of a canonical k-mer among n owners: LOW 32 bits of the hash, multiply-shift (the
// table's home slot uses the top bits of the same hash, kt_table.hpp)
__host__ __device__ __forceinline__ uint32_t owner_of(uint64_t kmer, uint32_t n) {
    return (uint32_t)(((khash(kmer) & 0xFFFFFFFFull) * (uint64_t)n) >> 32);
}

// reverse complement of a packed k-mer (2 bits/base) without a loop:
// complement, reverse the 2-bit groups of the 64-bit word, shift down.
__host__ __device__ __forceinline__ uint64_t rev_comp(uint64_t x, int k) {
    x = ~x;
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    x = (x >> 32) | (x << 32);
    return x >> (64 - 2 * k);
}

#ifndef KT_LBAR
#define KT_LBAR 1
#endif
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt == 0, i.e. for every
// global store just issued and every global load just prefetched by the wave - which is exactly what the
// kernels here want to leave in flight across the barrier.  Only valid where threads of a workgroup
// communicate through LDS alone (true for every kernel in this library).
__device__ __forceinline__ void lds_barrier() {
#if KT_LBAR
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#else
    __syncthreads();
#endif
}

// lane l receives the value of lane l-1; lane 0 receives `fill` (DPP wave_shr:1, gfx9)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false);
}

// inclusive prefix sum over the 64 lanes of a wave, DPP only (no LDS round trips): row_shr 1 / 2 / 4 / 8 inside the
// rows of 16 lanes, then row_bcast:15 (lane 15 of rows 0 and 2 into rows 1 and 3) and row_bcast:31 (lane 31 into rows 2, 3)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// c / d for integer-valued 0 <= c <= d < 2^32 given y = RN(1 / d): q0 = c*y, r = fma(-q0, d, c) (exact),
// q = fma(r, y, q0) = RN(c / d) (Markstein) - bit-identical to the reference's `vec[i] /= total`
// (composition/src/oligo.rs:255-257) at 4 f64 operations per bin instead of a ~14-operation division.
// kt_selftest_quotient checks every (c, d) a histogram can produce against the IEEE division.
__device__ __forceinline__ double quot_f64(double c, double d, double y) {
    const double q0 = __dmul_rn(c, y);
    return __fma_rn(__fma_rn(-q0, d, c), y, q0);
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// a * b + c with 32-bit factors and a 64-bit sum: one v_mad_u64_u32 (written with 64-bit operands the compiler makes two)
__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint32_t c) {
    return (uint64_t)a * (uint64_t)b + (uint64_t)c;
}

// A load through the SCALAR cache: p must be the same in every lane of the wave (the caller makes it so: uniform64) and
// point at memory nothing writes while the kernel runs.  Read as constant-address-space memory, which is what makes the
// compiler take the scalar path - a wave-uniform load of ordinary global memory stays a vector load whenever the
// kernel also stores (it cannot rule out a clobber), and a vector load's wait drains the wave's stores in flight.
template <class T>
__device__ __forceinline__ T load_uniform(const T *p) {
    typedef const T __attribute__((address_space(4))) *cptr;
    return *(cptr)(uintptr_t)p;
}

// uniform (scalar) broadcast of a 64-bit value held by all lanes
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

}  // namespace ktd
