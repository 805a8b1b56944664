// Internal (non-ABI) declarations shared by the translation units of libkmertools_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/kmertools_hip.h"

namespace kt {

void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define KT_HIP(expr)                                                                   \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            return kt::fail(KT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// growable device scratch owned by a ctx (only used by the KT_MEM_HOST staging path)
struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
};

constexpr int KT_MAX_OLIGO_K = 7;

}  // namespace kt

struct kt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cu = 256;
    uint32_t xcc_n = 0, xcc_map = 0;  // XCDs that run this device's workgroups, XCC_ID -> dense index in nibbles (kt_bulk.hip's census)
    // device copies of the canonical-bin LUT (u16[4^k], lut[f] = 4 * rank(min(f, rc f)): the byte offset of the
    // bin's u32 counter inside a row), k = 1..7
    uint16_t *lut_dev[kt::KT_MAX_OLIGO_K + 1] = {};
    uint32_t *lut32_dev[13] = {};  // canonical rank of every k-mer, k = 1..12 (the generic oligo path, kt_oligo_generic.hip)
    kt::Scratch s_bases, s_offsets, s_out, s_aux0, s_aux1, s_aux2;
    struct OligoKnobs {  // KT_OLIGO_* launch tunables, read once per context (kt_oligo.hip)
        bool loaded = false, live = false;
        uint32_t shape = 104, R = 0, oversub = 0, debug = 0, pw = 7;
        bool tune = true;
    } oligo_knobs;
    // Workgroups per resident slot of the k = 4 histogram launch, chosen by measurement per OUTPUT ARRAY (kt_oligo.hip,
    // oligo_launch): a few early large launches into an array cycle through the candidates with events around them,
    // the fastest stays for that array.  Which one is fastest goes with where the array lies in the memory, not with
    // the code (profiles/r3_oligo_placement.txt): the same process sees 1.90 and 2.40 ms on two of its allocations.
    struct OligoTune {
        static constexpr int NSET = 3, RING = 9, NEED = 3, WARM = 24, GIVE_UP = 45, ARRAYS = 8;
        struct Entry {                  // one output array (keyed by its address)
            const void *out = nullptr;
            uint64_t stamp = 0;         // last use (the least recently used entry makes room)
            bool decided = false;
            uint32_t pick = 96, launches = 0;
            double ns_per_read[NSET] = {0, 0, 0};  // sums while measuring, means once decided
            uint32_t kept[NSET] = {0, 0, 0};
        } arrays[ARRAYS];
        struct Trial { hipEvent_t a = nullptr, b = nullptr; const void *out = nullptr; uint32_t which = 0; uint64_t reads = 0; bool live = false; } ring[RING];
        uint64_t clock = 0;
        bool paused = false;            // kt_oligo_tuning(ctx, 0): launches neither count nor measure until switched on again
        uint32_t forced = 0;            // kt_oligo_tuning(ctx, n >= 2): paused, and every launch uses n workgroups per slot
        const void *last = nullptr;     // the array of the latest launch (kt_oligo_launch_info reports it)
        void release();
    } oligo_tune;
    int use();  // hipSetDevice
    int canon_lut(int k, const uint16_t **out);
    int canon_lut32(int k, const uint32_t **out);
};

// where level 2 of the partition passes reads a table's level-1 buckets from: the regions of buckets 0, 1, ... (cap1 keys of
// room each, `counts` keys used) of the level-1 output (kt_bulk.hip)
struct kt_seg_src {
    const void *keys;
    const uint64_t *counts;
    uint64_t cap1;
};

struct kt_bulk_job;  // kt_bulk.hip: the plan and buffers of a partition + range build in progress
void kt_bulk_job_free(kt_bulk_job *job);

struct kt_ctr {
    kt_ctx *ctx = nullptr;
    int k = 0;
    uint64_t cap = 0;          // m8 * 2^(n-3) slots (kttab::Geom)
    uint32_t shift = 0;        // 64 - n
    uint32_t m8 = 8;           // eighths of 2^n (slots per 4096-position range / 512)
    uint32_t kbits = 0;        // 2k when k <= 16: the table's hash is the bijection ktd::nhash (kttab::Geom)
    // the table of rank `owner` of a counter sharded over n_owners ranks (kt_shard.hip; a table of its own: 1, 0): a whole
    // table like any other - only kt_cov_batch_part looks at this (a k-mer that is not here may be on another rank)
    uint32_t n_owners = 1, owner = 0;
    // CUs the level-1 launches of a bulk job leave free (kt_shard.hip: the exchange's kernels run beside them on the comm stream,
    // and a level-1 workgroup takes a whole CU's LDS for the length of the launch)
    uint32_t l1_spare_cus = 0;
    // of the last bulk job into an export target: level-1 buckets whose keys did not fit their fixed fine regions and went
    // through the exact pass (of l2_buckets) - a caller that plans its jobs itself (kt_shard.hip) gives the next one more room
    uint32_t l2_redone = 0, l2_buckets = 0;
    bool paged_failed = false; // a bulk build overflowed a paged level-1 bucket: exact offsets from now on
    bool empty = true;         // nothing inserted since the last clear (bulk build allowed)
    bool needs_clear = true;   // slots hold stale data: clear before the incremental path / export
    bool dense = false;        // the last (bulk) build left every range packed, not as a probing image (kt_table.hpp)
    uint32_t *range_counts = nullptr;  // device, one per range: entries of the range while the table is dense
    // export target (kt_ctr_export_target): device arrays of the caller that a fresh bulk build writes its packed
    // entries to instead of the ranges' own slots - the build's output IS the export (no copy pass afterwards)
    uint64_t *xt_keys = nullptr;
    uint32_t *xt_counts = nullptr;
    uint64_t xt_max = 0;
    bool dense_ext = false;            // the table's entries ARE xt_keys / xt_counts [0, *distinct): (key, occurrences)
    const uint64_t *stage_keys = nullptr;  // kt_ctr_export_stage: where the staged entries are (device), how many
    const uint32_t *stage_counts = nullptr;
    uint64_t stage_n = 0;
    bool xt_too_small = false;         // the last build found the export target smaller than the table (reported by kt_bulk_finish)
    kt::Scratch b_ext;                 // the export-target build's holes and the scratch behind the caller's arrays
    kt::Scratch b_stage_k, b_stage_c;  // kt_ctr_export_stage's copy of the entries (when they are not the export target's arrays)
    kt::Scratch b_keys1, b_keys2, b_meta;  // bulk-build buffers (kt_bulk.hip), kept across calls
    kt::Scratch b_pack;                    // ... the reads packed for level 1 (PackedSource, KT_BULK_PACK)
    kt::Scratch b_desc;                    // ... and the descriptors of its record sources (kt_bulk_add_records)
    kt_bulk_job *job = nullptr;
    void *slots = nullptr;     // [cap] of {u64 key (KT_EMPTY_KEY = free), u32 count, u32 pad}
    uint32_t *flags = nullptr; // [0] = overflow flag, device
    uint64_t *cursor = nullptr; // device scalar for export
    uint64_t *distinct = nullptr; // device scalar: occupied slots (kept by every insert path, so kt_ctr_size is one 8-byte read)
};

// kt_oligo_generic.hip: histograms for k outside 3..7, counted in global memory (device pointers, enqueued)
int kt_oligo_generic_launch(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int k,
                            int count_min, int norm, int total_step, int dt, void *out);

namespace kt {
// host-side table builders (kt_host.cpp)
uint64_t rev_comp_bits(uint64_t kmer, int k);
// fills lut_full[4^k]: rank of the canonical form of every k-mer; returns kcount
uint32_t build_canon_lut(int k, uint16_t *lut_full);
}  // namespace kt
