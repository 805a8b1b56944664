// kt_table.hpp - the HBM-resident canonical k-mer table shared by the incremental (atomic)
// path in kt_ctr.hip, the lookups of kt_cov.hip and the bulk (partition + LDS build) path in kt_bulk.hip.
//
// Layout: 16-byte slots {u64 key, u32 occurrences-1, u32 pad}; KT_EMPTY_KEY marks a free slot.  The table is a
// row of independent RANGES: with x = the top n bits of khash(key), the top n - 13 bits of x select the range
// and the low 13 bits the home position inside it; a range has 1024 * m8 slots (m8 = 5..8 eighths of 8192, so that
// a table is at most 1.25x - not 2x - what was asked for) and is a closed linear-probing table of its own: probing
// runs forward from the home slot and wraps at the END OF THE RANGE, never into the next range.
//   cap = m8 * 2^(n-3) slots     home = (x >> 13) * (1024 * m8) + ((x & 8191) * m8 >> 3)
// Why ranges: "all keys of hash prefix p" is one contiguous, self-contained piece of the table, so the bulk path can
// build (or rebuild: merge a new batch into) every range in LDS on its own and write it with streaming stores - no
// key of one range ever lives in another, no clean-up pass through the atomic path (round 1's ranges spilled into
// their successors), and a full table is noticed after at most one trip round a range.  The price is one compare
// per probe step, and that a range - not the table - must have room: keep distinct keys below ~0.85 of the slots.
// Tables of fewer than 8192 slots are a single range of `cap` slots.
// The hash partitions of the out-of-core passes (ktd::owner_of) use the LOW 32 bits of the same hash, so a
// partition's keys still spread over the whole table.
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

#ifndef KT_LOG2_RANGE
#define KT_LOG2_RANGE 13
#endif
constexpr uint32_t LOG2_RANGE = KT_LOG2_RANGE, RANGE_FULL = 1u << LOG2_RANGE;

struct Geom {
    uint64_t cap;    // slots
    uint32_t shift;  // 64 - n, n = hash bits that address the table
    uint32_t m8;     // slots per range / 1024
    uint32_t kbits = 0;  // 2k for tables of k <= 16: the hash is ktd::nhash, a bijection of the 2k-bit k-mers (0: ktd::khash)
    // slots of one range (the whole table when it is smaller than a range)
    __host__ __device__ uint32_t range_slots() const {
        return shift > 64 - LOG2_RANGE ? (uint32_t)cap : m8 << (LOG2_RANGE - 3);
    }
};

inline Geom make_geom(uint64_t cap_request, int k = 0) {
    uint64_t p = 1024;
    uint32_t n = 10;
    while (p < cap_request) {
        p <<= 1;
        n++;
    }
    uint32_t m8 = 8;
    if (n >= LOG2_RANGE + 2)
        while (m8 > 5 && p / 8 * (m8 - 1) >= cap_request) m8--;
    return Geom{p / 8 * m8, 64 - n, m8, k >= 1 && k <= 16 ? 2u * (uint32_t)k : 0u};
}

struct TableRef {
    Slot *slots;
    Geom g;
    uint32_t *flags; // [0] = overflow flag
};

// a probe sequence: slot = base + rel, rel walks [0, rs) circularly from the key's home position
struct Probe {
    uint64_t base;
    uint32_t rel, rs;
    __host__ __device__ __forceinline__ uint64_t slot() const { return base + rel; }
    __host__ __device__ __forceinline__ void next() { rel = rel + 1 == rs ? 0 : rel + 1; }
};

__host__ __device__ __forceinline__ Probe probe_of(uint64_t key, const Geom &g) {
    const uint64_t x = ktd::khash_k(key, g.kbits) >> g.shift;  // n <= 54 bits
    if (g.shift > 64 - LOG2_RANGE) return Probe{0, (uint32_t)x, (uint32_t)g.cap};  // one small range, home = x
    const uint32_t rs = g.m8 << (LOG2_RANGE - 3);
    const uint64_t r = x >> LOG2_RANGE;
    return Probe{r * rs, (((uint32_t)x & (RANGE_FULL - 1)) * g.m8) >> 3, rs};
}

// table[key] += add; returns 0 = the key's range is full, 1 = the key was there,
// 2 = the key is new.
// A slot's key goes EMPTY -> key exactly once, so a stale (cached) probe can only show EMPTY for a slot that is now
// taken, and the CAS (device scope, coherent across XCDs) settles that case.  A k-mer seen once costs one probing load
// + one CAS (claiming the slot is its first count); a repeat costs one load + one 32-bit atomic add.  At most one
// trip round the key's range (<= 8192 probes), so a full table fails fast instead of being scanned end to end.
__device__ __forceinline__ uint32_t table_add(const TableRef &t, uint64_t key, uint32_t add) {
    Probe p = probe_of(key, t.g);
    for (uint32_t probe = 0; probe < p.rs; probe++) {
        Slot *s = t.slots + p.slot();
        uint64_t cur = __hip_atomic_load(&s->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == KT_EMPTY_KEY) {
            const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long *>(&s->key),
                                            (unsigned long long)KT_EMPTY_KEY, (unsigned long long)key);
            if (prev == KT_EMPTY_KEY) {
                if (add > 1u) atomicAdd(&s->count, add - 1u);
                return 2u;
            }
            cur = prev;
        }
        if (cur == key) {
            atomicAdd(&s->count, add);
            return 1u;
        }
        p.next();
    }
    return 0u;
}

}  // namespace kttab

// kt_bulk.hip: adds a whole read batch to the table without global atomics (partition by hash prefix, then every
// range is built - or rebuilt with what it already holds - in LDS).
// Returns KT_OK, or an error; `*done` = 0 when the batch / table shape is not eligible and
// the caller must use the incremental path.
// n_parts > 1: only the k-mers of hash partition `part` (ktd::owner_of(kmer, n_parts) == part) are counted
int kt_bulk_build(kt_ctr *ctr, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                  uint64_t total_bases, uint32_t n_parts, uint32_t part, int *done);
// same, from an array of canonical k-mers (one count each; KT_EMPTY_KEY entries are skipped)
int kt_bulk_build_keys(kt_ctr *ctr, const uint64_t *d_keys, uint64_t n_keys, int *done);
// The same build in phases, for batches that arrive in pieces (the sharded counter's records, kt_shard.hip):
// begin (plan + buffers for at most max_keys k-mers; *eligible = 0: use the probing path), level 1 over any number of
// sources (segments [seg_lo, seg_hi) of a read batch / arrays of canonical k-mers / records), finish (level 2 + range
// builds; one host round trip).  Sources must stay readable until finish.
int kt_bulk_begin(kt_ctr *ctr, uint64_t max_keys, int *eligible);
int kt_bulk_add_reads(kt_ctr *ctr, const uint8_t *d_bases, const uint64_t *d_offsets, const uint64_t *seg_first,
                      uint64_t n_reads, uint64_t n_seg, uint64_t seg_lo, uint64_t seg_hi, uint32_t n_parts, uint32_t part);
int kt_bulk_add_keys(kt_ctr *ctr, const uint64_t *d_keys, uint64_t n_keys);
// level 1 over records of at most 8 consecutive k-mers (kt_superkmer.hpp; what the ranks of the sharded counter send
// each other): n_runs stretches of whole blocks in device memory (`runs`: host array); kmers_bound = 0: the job was
// planned for everything these sources hold
namespace ktsk { struct RecRun; }
int kt_bulk_add_records(kt_ctr *ctr, const ktsk::RecRun *runs, uint32_t n_runs, uint64_t kmers_bound);
// ... and the same records through the probing path (one table_add per k-mer), outside any job
int kt_ctr_count_records(kt_ctr *ctr, const ktsk::RecRun *runs, uint32_t n_runs);
int kt_bulk_finish(kt_ctr *ctr);
// A fresh bulk build leaves the table DENSE: every range holds its entries packed at the front of its slots
// (kt_ctr.range_counts says how many) instead of a probing image - all that size / export need.  kt_table_image turns
// it into the probing layout in place (called by whatever has to probe: incremental adds, merges, cov);
// kt_table_dense_export writes the (key, count) pairs of a dense table to device arrays.
int kt_table_image(kt_ctr *ctr);
// kt_ctr.hip: an empty table filled from (key, occurrences) pairs on the device, *ctr->distinct of them (what
// kt_table_image does for a table whose entries live in the export target)
int kt_ctr_reload_pairs(kt_ctr *ctr, const uint64_t *d_keys, const uint32_t *d_counts);
int kt_table_dense_export(kt_ctr *ctr, uint64_t *d_keys, uint32_t *d_counts, uint64_t max_out, uint64_t *n);
