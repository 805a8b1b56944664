// kt_table.hpp - the HBM-resident canonical k-mer table shared by the incremental (atomic)
// path in kt_ctr.hip and the bulk (partition + LDS build) path in kt_bulk.hip.
//
// Layout: cap = 2^n or m * 2^(n-3) slots (m = 5, 6, 7) of 16 bytes {u64 key, u32 occurrences-1, u32 pad}; KT_EMPTY_KEY marks
// a free slot.  Home slot = TOP n bits of khash(key), linear probing forward (wrapping at
// cap).  Using the top bits makes "all keys of hash prefix p" one contiguous slot range, which
// is what lets the bulk path build the table range by range in LDS.  GPU ownership
// (ktd::owner_of) uses the LOW 32 bits of the same hash, so a shard's keys still spread over
// its whole table.
#pragma once
#include "kt_device.hpp"
#include "kt_internal.hpp"

namespace kttab {

struct Slot {
    uint64_t key;    // KT_EMPTY_KEY = free
    uint32_t count;  // occurrences - 1 (a claimed slot has been seen once)
    uint32_t pad;
};
static_assert(sizeof(Slot) == 16, "slot layout");

// Capacities come in eighths of a power of two, so that a table is at most 1.25x (not 2x) what was asked for.
// With x = the top n bits of the hash and m8 in 5..8:
//   cap = m8 * 2^(n-3)   home = (x >> 13) * (1024 * m8) + ((x & 8191) * m8 >> 3)
// i.e. every 8192-slot range of the 2^n layout shrinks to 1024 * m8 slots in place (m8 = 8: home = x).  The home
// slot is monotone in the hash and the top n - 13 bits of the hash select one contiguous range of slots - which is
// all the bulk build needs, so its partition passes are the same for every shape.
constexpr uint32_t LOG2_RANGE = 13, RANGE_FULL = 1u << LOG2_RANGE;

struct Geom {
    uint64_t cap;
    uint32_t shift;  // 64 - n
    uint32_t m8;     // slots per range / 1024
    __host__ __device__ uint32_t range_slots() const { return m8 << (LOG2_RANGE - 3); }
};

inline Geom make_geom(uint64_t cap_request) {
    uint64_t p = 1024;
    uint32_t n = 10;
    while (p < cap_request) {
        p <<= 1;
        n++;
    }
    uint32_t m8 = 8;
    if (n >= LOG2_RANGE + 2)
        while (m8 > 5 && p / 8 * (m8 - 1) >= cap_request) m8--;
    return Geom{p / 8 * m8, 64 - n, m8};
}

struct TableRef {
    Slot *slots;
    Geom g;
    uint32_t *flags; // [0] = overflow flag
};

__host__ __device__ __forceinline__ uint64_t home_slot(uint64_t key, const Geom &g) {
    const uint64_t x = ktd::khash(key) >> g.shift;  // n <= 54 bits
    if (g.m8 == 8) return x;
    return (x >> LOG2_RANGE) * g.range_slots() + ((((uint32_t)x & (RANGE_FULL - 1)) * g.m8) >> 3);
}
__host__ __device__ __forceinline__ uint64_t next_slot(uint64_t slot, const Geom &g) {
    return slot + 1 == g.cap ? 0 : slot + 1;
}

// table[key] += add.  A slot's key goes EMPTY -> key exactly once, so a stale (cached) probe
// can only show EMPTY for a slot that is now taken, and the CAS (device scope, coherent across
// XCDs) settles that case.  A k-mer seen once costs one probing load + one CAS (claiming the
// slot is its first count); a repeat costs one load + one 32-bit atomic add.
__device__ __forceinline__ bool table_add(const TableRef &t, uint64_t key, uint32_t add) {
    uint64_t slot = home_slot(key, t.g);
    for (uint64_t probe = 0; probe < t.g.cap; probe++) {
        uint64_t cur = __hip_atomic_load(&t.slots[slot].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == KT_EMPTY_KEY) {
            const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long *>(&t.slots[slot].key),
                                            (unsigned long long)KT_EMPTY_KEY, (unsigned long long)key);
            if (prev == KT_EMPTY_KEY) {
                if (add > 1u) atomicAdd(&t.slots[slot].count, add - 1u);
                return true;
            }
            cur = prev;
        }
        if (cur == key) {
            atomicAdd(&t.slots[slot].count, add);
            return true;
        }
        slot = next_slot(slot, t.g);
    }
    return false;
}

}  // namespace kttab

// kt_bulk.hip: builds an EMPTY table from a whole read batch without global atomics.
// Returns KT_OK, or an error; `*done` = 0 when the batch / table shape is not eligible and
// the caller must use the incremental path.
int kt_bulk_build(kt_ctr *ctr, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                  uint64_t total_bases, int *done);
// same, from an array of canonical k-mers (one count each; KT_EMPTY_KEY entries are skipped)
int kt_bulk_build_keys(kt_ctr *ctr, const uint64_t *d_keys, uint64_t n_keys, int *done);
