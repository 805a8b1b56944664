// kt_superkmer.hpp - what the sharded counter (kt_shard.hip) puts on the wires, and who owns a k-mer.
//
// The reference partitions its table by `min_mer % n_parts` (counter/src/lib.rs:100,127) - any function of the canonical
// k-mer would do, the partition is not observable in the output.  Here the owner of a k-mer is a function of its
// MINIMISER: of the w = k - m + 1 m-mers inside the k-mer, each taken in canonical form (the smaller of the m-mer and its
// reverse complement, kmer/src/minimiser.rs:61-175 orders by that value; here the order is a hash of it so that the
// owners are balanced), the one with the smallest hash h.  owner = top bits of mix(min h) scaled to n_owners.  A k-mer and its reverse
// complement hold the same canonical m-mers, so both strands of a k-mer have one owner; consecutive k-mers of a read
// mostly share their minimiser, so a read falls into a few RUNS of k-mers with one owner each (a "super-k-mer": n
// k-mers in n + k - 1 bases), and what travels is bases at 2 bits, not k-mers at 8 bytes.
//
// Wire format: RECORDS of at most 8 consecutive k-mers of one owner (a run longer than 8 k-mers is cut every 8: a
// record is what one lane of the receiver's level-1 pass takes per round - kt_bulk.hip, RecordSource - so the lanes stay
// 3/4 full, where the reads themselves keep them 4/5 full).  A record is 80 bits:
//     a   u64   bases 0..31 of the record, 2 bits each, the first base in the top bits
//     b   u16   bits 15..4: bases 32..37;  bits 3..0: the number of k-mers, 0..8 (0: no record)
// (8 + k - 1 <= 38 bases; what lies behind the last k-mer's last base is zero.)  Records are kept in BLOCKS of 1024:
// 1024 x a, then 1024 x b - 10 240 bytes, 1280 u64 words.  An owner's region of the send buffer is a row of blocks; only
// the last block of a region is partly filled.
#pragma once
#include <stdint.h>

#include "kt_device.hpp"

namespace ktsk {

constexpr uint32_t REC_KMERS = 8;        // k-mers per record at most
constexpr uint32_t BLOCK_RECS = 1024;    // records per block
constexpr uint32_t BLOCK_WORDS = 1280;   // u64 words per block: 1024 a + 256 words of b
constexpr uint32_t BLOCK_BYTES = BLOCK_WORDS * 8;
constexpr uint32_t MAX_OWNERS = 64;

// the minimiser length for k-mers of length k: the window w = k - m + 1 is the largest power of two <= 16 that leaves
// m >= 8 (a power of two: the sliding minimum is log2(w) passes; m <= 16: a canonical m-mer is one 32-bit word); k <= 8:
// the k-mer is its own minimiser
__host__ __device__ inline uint32_t window_of(uint32_t k) {
    uint32_t w = 16;
    while (w > 1 && k < w + 7) w >>= 1;
    return w;
}
__host__ __device__ inline uint32_t mmer_of(uint32_t k) { return k - window_of(k) + 1; }

// (a[23:0] * b[23:0]) mod 2^32: one full-rate instruction on the device (v_mul_u32_u24; a 32-bit multiply takes four)
__host__ __device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu));
#endif
}
// the order of the canonical m-mers: a 32-bit mix of the (at most 32-bit) m-mer - its low 24 bits and its bits 12 .. 31
// through one 24-bit multiply each (three instructions; not a bijection: two m-mers may tie - the owner is a function of
// the smallest HASH, whatever m-mer it came from.  Run lengths and balance measured equal to murmur3's finaliser.)
__host__ __device__ __forceinline__ uint32_t mhash(uint32_t x) { return mul24(x, 0x9E3779u) + mul24(x >> 12, 0x85EBCBu); }
// the owner from the smallest hash of the window (small values: mixed once more before the top bits are taken)
__host__ __device__ __forceinline__ uint32_t owner_of_min(uint32_t hmin, uint32_t n_owners) {
    uint32_t x = mul24(hmin, 0x9E3779u) + mul24(hmin >> 8, 0x85EBCBu);
    x ^= x >> 16;
    return mul24(x >> 8, n_owners) >> 24;  // (n_owners <= 64)
}

// reverse complement of an m-mer held in the low 2m bits of a 32-bit word (m <= 16)
__host__ __device__ __forceinline__ uint32_t rev_comp32(uint32_t x, uint32_t m) {
    x = ~x;
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    x = (x >> 16) | (x << 16);
    return x >> (32u - 2u * m);
}

// owner of a k-mer (either strand) among n_owners: the definition, one k-mer at a time (host helpers, the finalize
// rounds' pending pairs; the route kernel computes the same thing with a rolling window)
__host__ __device__ inline uint32_t owner_of_kmer(uint64_t kmer, uint32_t k, uint32_t n_owners) {
    if (n_owners <= 1) return 0;
    const uint32_t m = mmer_of(k), w = k - m + 1;
    const uint32_t mask = m == 16 ? 0xFFFFFFFFu : (1u << (2u * m)) - 1u;
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < w; i++) {
        const uint32_t f = (uint32_t)(kmer >> (2u * (k - m - i))) & mask;  // the m-mer that starts at base i
        const uint32_t r = rev_comp32(f, m);
        const uint32_t h = mhash(f < r ? f : r);
        best = h < best ? h : best;
    }
    return owner_of_min(best, n_owners);
}

// one stretch of whole blocks of records: `n_rec` records in ceil(n_rec / 1024) blocks at `blocks`
struct RecRun {
    const uint64_t *blocks;
    uint64_t n_rec;
};

}  // namespace ktsk
