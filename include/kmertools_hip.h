/*
 * kmertools_hip.h - C ABI of libkmertools_hip.so, the MI355X (gfx950) drop-in for
 * kmertools' k-mer hot path.
 *
 * The reference (anuradhawick/kmertools, pure Rust) has no FFI today; the entry
 * points below are what a `extern "C"` block in the reference's crates would
 * bind in place of the Rust functions cited on each declaration (paths relative
 * to the reference root).  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *  - plain pointers and sizes only; the caller owns every input and output buffer;
 *    the library owns device scratch inside kt_ctx / kt_ctr.
 *  - every function returns KT_OK (0) or a KT_ERR_* code; kt_last_error() gives the
 *    message for the calling thread.  Nothing aborts or throws across the ABI.
 *  - read batches are CSR: `bases` = concatenated ASCII bytes (no separators),
 *    `offsets` = uint64[n_reads + 1], read i = bases[offsets[i] .. offsets[i+1]).
 *  - `mem` says where `bases`, `offsets` and the output buffers live:
 *    KT_MEM_DEVICE = all are device (HBM) pointers, work is enqueued on the ctx
 *    stream and the call returns without waiting for it to finish (a few calls
 *    say that they synchronise; a large batch into an empty k-mer table waits
 *    once or twice for a 4-byte flag in the middle of its kernels);
 *    KT_MEM_HOST   = all are host pointers, the library stages them through its
 *    own device scratch and returns after the results are back on the host.
 *  - k-mers are uint64 (`type Kmer = u64`, kmer/src/lib.rs:4), 2 bits per base,
 *    A=0 C=1 G=2 T=3, first base most significant.
 */
#ifndef KMERTOOLS_HIP_H
#define KMERTOOLS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KT_OK 0
#define KT_ERR_ARG 1      /* bad argument (k out of range, null pointer, ...) */
#define KT_ERR_HIP 2      /* a HIP runtime call failed; message has the HIP error */
#define KT_ERR_NOMEM 3    /* device or host allocation failed */
#define KT_ERR_FULL 4     /* k-mer table ran out of slots (raise capacity) */
#define KT_ERR_NODEVICE 5 /* no usable gfx950 device */
#define KT_ERR_BADNT 6    /* kt_cgr_points: a byte outside ACGTUacgtu ("Bad nucleotide, unable to proceed") */

#define KT_MEM_HOST 0
#define KT_MEM_DEVICE 1

#define KT_F64 0 /* reference's element type (Vec<f64>) */
#define KT_F32 1 /* BASELINE cfg5 output type */
#define KT_U32 2 /* raw integer counts (norm must be 0) */

#define KT_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull /* never a k-mer: k <= 31 => key < 2^62 */

typedef struct kt_ctx kt_ctx; /* one device + one stream; use from one host thread at a time */
typedef struct kt_ctr kt_ctr; /* HBM-resident canonical k-mer count table */

/* ---- library ----------------------------------------------------------------- */
int kt_version(void);
const char *kt_last_error(void);
int kt_device_count(int *count);

/* own_stream == 0: enqueue on `stream`, a hipStream_t of the caller (e.g. torch's current
 * stream; NULL is the device's default stream).  own_stream != 0: `stream` is ignored and
 * the ctx creates and owns a private non-blocking stream. */
int kt_ctx_create(int device, void *stream, int own_stream, kt_ctx **out);
int kt_ctx_destroy(kt_ctx *ctx);
int kt_ctx_sync(kt_ctx *ctx);
/* free / total HBM of the context's device in bytes (hipMemGetInfo): lets a caller size a table to what fits */
int kt_device_memory(kt_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);

/* A device array that is fast to write.  Where a large allocation lands in the HBM moves the store-bound kernels by up
 * to 20 % - reproducibly per allocation, with no difference in plain fill rate - so a caller that keeps
 * an output (or input) array for many launches can have the library choose among `candidates` allocations of `bytes`:
 * all are allocated (fewer if they would take more than 60 % of the free memory), `probe(user, array)` - the caller's
 * function, which enqueues the work the array is meant for on the context's stream, e.g. kt_oligo_batch into it - runs
 * 20 times on the first one and then `launches` times on each between two events (its first call into an array is not
 * counted), the fastest array is returned in *out and the others are freed.  ms[i] (may be NULL; room for `candidates`
 * values) = milliseconds per probe on candidate i; candidate 0 is what a plain allocation would have been; *n_tried,
 * *picked may be NULL.  probe == NULL or candidates <= 1: a plain allocation.  Release with kt_device_free.
 * No reference counterpart: the Rust side allocates its Vecs where the allocator puts them (composition/src/oligo.rs:147). */
typedef int (*kt_probe_fn)(void *user, void *candidate_dev);
int kt_device_alloc_placed(kt_ctx *ctx, uint64_t bytes, int candidates, int launches, kt_probe_fn probe, void *user,
                           void **out, double *ms, int *n_tried, int *picked);
int kt_device_free(kt_ctx *ctx, void *ptr);

/* Page-locks / releases a host buffer of the caller (hipHostRegister) so that KT_MEM_HOST calls move it
 * by DMA at full PCIe rate instead of through the driver's pageable staging.  Optional: every entry point
 * accepts ordinary pageable memory.  Worth it for buffers that are reused over many batches (the
 * reference's batch loops, composition/src/oligo.rs:147-164, reuse theirs the same way). */
int kt_host_register(kt_ctx *ctx, void *ptr, size_t bytes);
int kt_host_unregister(kt_ctx *ctx, void *ptr);

/* ---- host-side helpers (no GPU work) ------------------------------------------ */

/* number of output bins: canonical count (count_min != 0) or 4^k.
 * replaces: `kcount` from KmerGenerator::kmer_pos_maps, kmer/src/kmer.rs:54-73;
 *           the raw-mode size in composition/src/oligo.rs:232-236 */
int kt_bins(int k, int count_min, uint64_t *bins);

/* replaces: KmerGenerator::kmer_pos_maps, kmer/src/kmer.rs:54-73.
 * min_mer_pos_map[4^k]: rank of canonical k-mer among canonicals (0 in non-canonical
 * slots, as in the reference); pos_min_mer[kcount]: rank -> canonical k-mer (the
 * reference's HashMap<pos,kmer> as a dense array).  Either may be NULL. */
int kt_pos_map(int k, uint32_t *min_mer_pos_map, uint64_t *pos_min_mer, uint32_t *kcount);

/* replaces: KmerGenerator::rev_comp, kmer/src/kmer.rs:43-52 */
uint64_t kt_rev_comp(uint64_t kmer, int k);

/* replaces: numeric_to_kmer, kmer/src/lib.rs:19-34 (out: k+1 bytes, NUL-terminated) */
int kt_numeric_to_kmer(uint64_t kmer, int k, char *out);

/* replaces: kmer_to_numeric, kmer/src/lib.rs:36-50 (no validity check, like the
 * reference; len > 32 is KT_ERR_ARG = the ValueError of pybindings/src/kmer.rs:58-63) */
int kt_kmer_to_numeric(const char *kmer, uint64_t len, uint64_t *fwd, uint64_t *rev);

/* replaces: OligoCgrComputer::cgr_maps + the per-k-mer midpoint walk,
 * composition/src/oligocgr.rs:165-189, :123-143.  xy[2*i], xy[2*i+1] = CGR point of
 * canonical k-mer i; read-independent, so computed once on the host. */
int kt_cgr_coords(int k, double vecsize, double *xy);

/* ---- device hot path ----------------------------------------------------------- */

/* replaces: KmerGenerator::new + Iterator::next, kmer/src/kmer.rs:30-41, :80-106
 * (and pybindings/src/kmer.rs:22-41).  Position-parallel form: for every base index
 * g in [0, offsets[n_reads]) valid[g] = 1 iff a k-mer ends at that base (its k bases
 * lie in one read and are all ACGTU/acgtu/0..3), and then fwd[g], rev[g] hold the
 * pair the reference's iterator yields at that position.  Iterating g in order and
 * skipping valid[g]==0 reproduces the iterator.  1 <= k <= 31. */
int kt_kmers(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
             int k, uint64_t *fwd, uint64_t *rev, uint8_t *valid, int mem);

/* replaces: OligoComputer::vectorise_one, composition/src/oligo.rs:231-259;
 *           OligoCgrComputer::seq_to_kmer, composition/src/oligocgr.rs:145-163;
 *           python OligoComputer.vectorise_one/_batch, pybindings/src/oligo.rs:39-81.
 * out: n_reads x bins row-major (bins from kt_bins), element type out_dtype.
 * count_min: merge reverse complements (canonical bins) / 0 = raw 4^k bins.
 * norm: divide every bin by max(1, total), total = total_step * (#k-mers of the read).
 * total_step: 1 (CLI crate, oligo.rs:248,251) or 2 (python raw mode, pybindings
 * oligo.rs:61).  KT_F64 results are bit-identical to the reference (integer counts,
 * one IEEE division).  1 <= k <= 12: 3..7 (the CLI range, kmertools/src/args.rs:85) is the LDS kernel; the Python
 * class takes any k (pybindings/src/oligo.rs:22-31), so the other values count in global memory - correct, not
 * fast - up to the k whose 4^k-entry rank map the reference itself could still allocate comfortably. */
int kt_oligo_batch(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                   int k, int count_min, int norm, int total_step, int out_dtype, void *out,
                   int mem);

/* No counterpart in the reference (a launch detail made visible): the number of workgroups per resident slot the k = 4
 * histogram launches into the output array of the latest launch use.  It is chosen by measurement, per output array
 * (which setting is fastest goes with where the array lies in the memory: DESIGN.md 4.1): a few early large launches
 * (>= ~6 M reads of 150 bases) into an array cycle through 32 / 96 / 200 with events around them, the fastest stays.
 * *decided = 0 while still measuring (then *wgs_per_slot is the default, 96, and the means are 0); ns_per_read[3] =
 * the means for 32, 96, 200 in ns per read.  Results never depend on it.  KT_OLIGO_TUNE=0 in the environment keeps
 * the default, KT_OLIGO_OVERSUB=n fixes n.  (The trial launches record a pair of events on the context's stream.  A
 * launch made while that stream is being captured into a graph records and queries nothing - it runs with what has
 * been decided for its array, or the default, like a launch after kt_oligo_tuning(ctx, 0) - and an event that cannot be
 * created or recorded drops the trial, never the call.) */
int kt_oligo_launch_info(kt_ctx *ctx, uint32_t *wgs_per_slot, int *decided, double *ns_per_read);
/* mode 0: the launches that follow neither count towards nor take part in that measurement (they use what has been
 * decided for their array, else the default) - for a caller that is timing launches itself; mode 1 resumes;
 * mode n >= 2: as 0, and every k = 4 launch uses n workgroups per resident slot (a caller comparing shapes itself). */
int kt_oligo_tuning(kt_ctx *ctx, int mode);

/* Self-test of the f64 normalisation: the kernels compute `vec[i] /= max(1, total)` (composition/src/oligo.rs:255-257)
 * as a reciprocal + two fused multiply-adds per bin instead of a division.  For every divisor d in [d_lo, d_hi]
 * and every count c in 0..d this runs that exact device code and the IEEE division side by side:
 * *n_checked pairs, *n_mismatch of them with different bits (must be 0), *checksum = sum of the division's
 * bit patterns mod 2^64 (lets a host check the device's division itself).  Synchronises. */
int kt_selftest_quotient(kt_ctx *ctx, uint32_t d_lo, uint32_t d_hi, uint64_t *n_checked, uint64_t *n_mismatch,
                         uint64_t *checksum);

/* replaces: CountComputer::new / count_chunk's table, counter/src/lib.rs:37-55, :100.
 * One HBM-resident open-addressing table (u64 keys, u32 counts - the reference's
 * types) takes the place of the reference's n_parts scc maps and chunk files.
 * capacity_slots is rounded up to the next m * 2^j with m in 5..8 (at most 1.25x the request; at least 1024);
 * keep distinct keys <= ~70 % of it. */
int kt_ctr_create(kt_ctx *ctx, int k, uint64_t capacity_slots, kt_ctr **out);
int kt_ctr_destroy(kt_ctr *ctr);
int kt_ctr_clear(kt_ctr *ctr);
/* the slots the table really has (capacity_slots after rounding) */
int kt_ctr_capacity(kt_ctr *ctr, uint64_t *slots);

/* replaces: the hot loop of count_chunk, counter/src/lib.rs:119-131:
 * for every k-mer of every read: table[min(fwd,rev)] += 1.  Repeatable (= chunks). */
int kt_ctr_add_reads(kt_ctr *ctr, const uint8_t *bases, const uint64_t *offsets,
                     uint64_t n_reads, int mem);

/* The same, restricted to one of n_parts hash partitions: only k-mers with kt_owner_of(kmer, n_parts) == part are
 * counted.  replaces: the reference's answer to "the k-mers do not fit memory" - it cuts the input into chunks, spills
 * every chunk's partitions (`min_mer % n_parts`) to temp files and merges partition by partition
 * (counter/src/lib.rs:114-118, :127, :151-167, :188-231).  Here a table that cannot hold every distinct k-mer is
 * filled n_parts times, pass p with partition p of the whole input, exported and cleared in between: the union of
 * the exports is the complete answer (every k-mer lives in exactly one partition, with all of its occurrences). */
int kt_ctr_add_reads_part(kt_ctr *ctr, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int mem,
                          uint32_t n_parts, uint32_t part);

/* replaces: merge's arithmetic, counter/src/lib.rs:201-210: table[keys[i]] += counts[i]
 * (counts == NULL means 1 each: raw canonical k-mers routed from another GPU). */
int kt_ctr_add_pairs(kt_ctr *ctr, const uint64_t *keys, const uint32_t *counts, uint64_t n,
                     int mem);

/* number of distinct k-mers in the table (synchronises). KT_ERR_FULL if an insert overflowed. */
int kt_ctr_size(kt_ctr *ctr, uint64_t *distinct);

/* replaces: map.scan, counter/src/lib.rs:162-165, :220-230.  Writes up to max_out
 * (key,count) pairs in unspecified order (the reference's order is unspecified too,
 * its tests sort); *n_out = number written.  Synchronises. */
int kt_ctr_export(kt_ctr *ctr, uint64_t *keys, uint32_t *counts, uint64_t max_out,
                  uint64_t *n_out, int mem);

/* The same map.scan (counter/src/lib.rs:162-165, :220-230) in pieces, for callers that must not hold a table of billions
 * of entries in host memory at once - the reference streams its scan straight into the output file (:220-230).
 * kt_ctr_export_stage gathers the entries on the DEVICE (a staging area of the library; a table counted into an export
 * target has them there already) and returns their number; kt_ctr_export_fetch then copies entries
 * [first, first + count) to host arrays, any number of times, in any order.  The staged entries stay valid until the table
 * is changed (kt_ctr_clear, adds) or another export / cov call uses the context's scratch. */
int kt_ctr_export_stage(kt_ctr *ctr, uint64_t *n_out);
int kt_ctr_export_fetch(kt_ctr *ctr, uint64_t first, uint64_t count, uint64_t *keys_host, uint32_t *counts_host);

/* Where the table's entries are wanted - told BEFORE counting, so that counting can deliver them there.
 * replaces: the same map.scan as kt_ctr_export (counter/src/lib.rs:162-165, :220-230), for the usual life of a
 * table: filled once, written out once.  keys_dev / counts_dev are DEVICE arrays of max_out entries owned by the
 * caller.  From now on (sticky; NULL, NULL, 0 switches it off) every whole-batch count into an EMPTY table
 * (kt_ctr_add_reads / kt_sharded_add_reads after creation or kt_ctr_clear) writes its (key, count) pairs straight
 * into these arrays as the last step of the build, and kt_ctr_size / kt_ctr_export(keys_dev, counts_dev, ...) then
 * only report the number of entries - no second pass over the table.  The table keeps referring to the arrays
 * (entries [0, size) in unspecified order; what lies beyond them in the arrays is unspecified) until the next call that changes or probes it (a further add, kt_cov_batch,
 * an export elsewhere, a new target), which first rebuilds its own probing copy from them: leave them untouched
 * until then, or call kt_ctr_clear.  If the arrays turn out too small, the call that counted returns KT_ERR_ARG AFTER
 * having built the table in its own slots: nothing is lost - kt_ctr_size, kt_ctr_export into arrays of that size, further
 * adds all work - and the target arrays hold nothing usable (they are never written past max_out). */
int kt_ctr_export_target(kt_ctr *ctr, uint64_t *keys_dev, uint32_t *counts_dev, uint64_t max_out);

/* replaces: CgrComputer::vectorise_one, composition/src/cgr.rs:127-144 (corners from cgr_maps,
 * :12-36) and python CgrComputer.vectorise_one/_batch, pybindings/src/cgr.rs:38-62.
 * Whole-sequence chaos game walk: xy[2*g], xy[2*g+1] = marker after base g of the batch
 * (g = global base index, so read i owns xy[2*offsets[i] .. 2*offsets[i+1])); every read
 * starts from (vecsize/2, vecsize/2).  Bit-identical to the reference's serial f64 walk.
 * A byte outside ACGTUacgtu is an error for the whole call, as in the reference:
 * *bad_pos (may be NULL) = lowest offending base index, UINT64_MAX if none.  KT_MEM_HOST:
 * returns KT_ERR_BADNT in that case; KT_MEM_DEVICE: bad_pos is a device pointer the caller
 * inspects after synchronising (the call itself returns KT_OK). */
int kt_cgr_points(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                  double vecsize, double *xy, uint64_t *bad_pos, int mem);

/* replaces: MinimiserGenerator::new + Iterator::next, kmer/src/minimiser.rs:36-56, :61-175 (and
 * python MinimiserGenerator, pybindings/src/min.rs:24-47; the per-record loops of
 * misc/src/minimisers.rs:43-54, :124-135).  For every read, in iterator order, the triples
 * (minimiser, window start, window end) the reference yields; read i owns entries
 * [ev_offsets[i], ev_offsets[i+1]) of kmers/starts/ends (starts/ends are read-local, ends
 * exclusive).  wsize = 0 means "the read's own length" (one minimiser per read,
 * misc/src/minimisers.rs:44-48); otherwise msize <= wsize (windows of more than 4096 m-mers: a two-level
 * sliding minimum in front of the same kernel, three scratch arrays of one m-mer per base; windows of 2^30 bases and
 * more take a one-read-per-thread path: correct, and parallel across reads only).
 * 1 <= msize <= 31.  Quirks of the iterator are kept (a change on the last base swallows the
 * final window; a last run shorter than the window reports UINT64_MAX).  Reads shorter than msize
 * yield nothing with wsize = 0 (the reference's capacity arithmetic underflows there).
 * Always synchronises.  *n_events = number of triples.  capacity = room in kmers/starts/ends:
 * 0 counts only (ev_offsets is then unspecified); if 0 < capacity < *n_events the first
 * `capacity` triples are written and KT_ERR_ARG is returned. */
int kt_minimisers(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                  uint64_t wsize, int msize, uint64_t *ev_offsets, uint64_t *kmers,
                  uint64_t *starts, uint64_t *ends, uint64_t capacity, uint64_t *n_events, int mem);

/* replaces: CovComputer::vectorise_one, coverage/src/lib.rs:165-184 (and the HashMap
 * re-load of kmers.counts at :82-92: the table is probed where kt_ctr_add_reads left it).
 * For every canonical k-mer of a read (k = the table's k): count = table[kmer] or 0,
 * bin = min(count / bin_size, bin_count - 1); out row = histogram of the bins, divided
 * by max(1, #k-mers of the read) when norm != 0.  out: n_reads x bin_count row-major,
 * element type out_dtype (KT_U32 needs norm = 0).  KT_F64 results are bit-identical to
 * the reference (integer counts, one IEEE division).  bin_size, bin_count >= 1. */
int kt_cov_batch(kt_ctr *table, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                 uint64_t bin_size, uint64_t bin_count, int norm, int out_dtype, void *out,
                 int mem);

/* counts[i] = occurrences of the canonical k-mer keys[i] in the table, 0 when it is absent (`mem` says where keys and
 * counts live).  replaces: the HashMap get of coverage/src/lib.rs:170 (`kmer_counts.get(&kmer)`), one key at a time. */
int kt_ctr_lookup(kt_ctr *table, const uint64_t *keys, uint64_t n, uint32_t *counts, int mem);

/* The same lookups against a table that holds only PART of the k-mers - one hash partition of an out-of-core count
 * (kt_ctr_add_reads_part: n_parts passes, the table refilled for each) or one shard of a sharded table (n_parts = 1).
 * Only the k-mers this table answers for are binned (a k-mer of another partition is not "absent": it is skipped), as
 * raw u32 counts ADDED to `counts` (n_reads x bin_count, zeroed by the caller before the first part; `mem` says where
 * it lives).  Summed over the parts - modulo 2^32, cell by cell - every k-mer of every read has been binned exactly
 * once: the rows kt_cov_batch gives with KT_U32; normalisation (count / max(1, sum of the row),
 * coverage/src/lib.rs:180-182) is then one division per cell.  (A shard does not know which k-mers the other shards
 * hold without their minimisers: shard 0's pass puts every k-mer into bin 0, and the shard that holds a k-mer moves it
 * from there to its bin - a shard's own rows mean nothing before they are summed.)
 * replaces: coverage/src/lib.rs:69-92 + :165-184 for inputs whose k-mers do not fit one table. */
int kt_cov_batch_part(kt_ctr *table, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                      uint64_t bin_size, uint64_t bin_count, uint32_t *counts, int mem, uint32_t n_parts,
                      uint32_t part);

/* Multi-GPU routing step (the reference's `min_mer % n_parts` partitioning,
 * counter/src/lib.rs:127, re-expressed as hash-prefix ownership):
 * writes every canonical k-mer of the reads into keys_out grouped by owner
 * o = kt_owner_of(kmer, n_owners); owner_counts[o] = group sizes (groups are
 * contiguous, in owner order).  keys_out needs room for offsets[n_reads] entries
 * (upper bound).  owner_counts follows `mem`.  1 <= n_owners <= 64. */
int kt_ctr_route(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                 int k, int n_owners, uint64_t *keys_out, uint64_t *owner_counts, int mem);

/* ---- k-mer counting sharded over the GPUs of a node (ownership by hash prefix of the k-mer's minimiser) -----------
 * replaces: the `n_parts` partitioning of counter/src/lib.rs:100,127,243-247 and the per-partition merge of
 * :188-231 - the partitions are GPUs.  Rank o (one process or thread, one kt_ctx, one GPU) counts the k-mers whose
 * MINIMISER - the canonical m-mer inside the k-mer with the smallest hash (kmer/src/minimiser.rs:61-175 defines a
 * minimiser; m = kt_shard_minimiser(k)) - hashes into its share of the hash range: owner = mix(min hash) * N >> 32,
 * kt_sharded_owner_of / kt_shard_owner_of.  Both strands of a k-mer hold the same canonical m-mers, and consecutive
 * k-mers of a read mostly share their minimiser: a read falls into a few runs of k-mers with one owner each, and what
 * the ranks send each other is the runs' bases at 2 bits (records of at most 8 k-mers in 10 bytes: ~1.7 bytes per k-mer
 * at k = 31), not the k-mers at 8 bytes.  Every rank's shard is a whole table of its own (kt_sharded_table).
 * kt_sharded_add_reads and kt_sharded_finalize are COLLECTIVE: every rank calls them the same number of times (a rank
 * without reads passes n_reads = 0).  Each rank routes its own reads into one region of records per owner, the ranks
 * tell each other the regions' sizes, the regions travel in pieces (grouped ncclSend / ncclRecv from librccl over xGMI,
 * or the caller's host all-to-all) while the partition passes of the ordinary single-GPU pipeline already run over what
 * has arrived.  Results stay sharded: the union of the ranks' exports is the answer (the reference's output order is
 * unspecified anyway).  With n_ranks == 1 everything degenerates to kt_ctr_add_reads.  A rank that cannot take part in
 * a collective call (a batch larger than agreed, a pending table that overflowed) still enters the exchange that
 * carries the ranks' status, and EVERY rank returns an error - none is left waiting. */
typedef struct kt_sharded kt_sharded;

/* 128-byte RCCL unique id (ncclGetUniqueId): rank 0 makes it, the caller hands it to the other ranks (any channel:
 * MPI, a file, torch.distributed's store) */
int kt_rccl_unique_id(uint8_t *id128);

/* capacity_slots: slots of this rank's shard (a table of its own: size it for the k-mers one rank will own - about
 * 1 / n_ranks of the distinct k-mers, more where a few minimisers dominate); max_batch_bases: the largest batch any
 * rank will pass to kt_sharded_add_reads - it fixes the room of the exchange regions, so it must be the same on every
 * rank (a mismatch is reported by every rank at the first kt_sharded_add_reads). */
int kt_sharded_create_rccl(kt_ctx *ctx, int k, uint64_t capacity_slots, uint64_t max_batch_bases, int n_ranks, int rank,
                           const uint8_t *id128, kt_sharded **out);

/* host transport: `fn(user, send, recv, bytes_per_rank)` must move block p of `send` (host memory, n_ranks blocks of
 * bytes_per_rank) to rank p and fill block p of `recv` with what rank p sent to this rank; returns 0 on success.
 * The library stages the device regions through page-locked host memory around the call. */
typedef int (*kt_alltoall_fn)(void *user, const void *send, void *recv, uint64_t bytes_per_rank);
int kt_sharded_create_host(kt_ctx *ctx, int k, uint64_t capacity_slots, uint64_t max_batch_bases, int n_ranks, int rank,
                           kt_alltoall_fn fn, void *user, kt_sharded **out);
/* The same in two steps, for callers that want to agree between them: kt_sharded_create_local allocates everything
 * (the shard, the exchange buffers) and needs no peer; once EVERY rank has succeeded (the caller's own barrier: a
 * rank whose allocation failed never enters ncclCommInitRank, where its peers would wait for it), kt_sharded_connect_rccl
 * / kt_sharded_connect_host brings the transport up. */
int kt_sharded_create_local(kt_ctx *ctx, int k, uint64_t capacity_slots, uint64_t max_batch_bases, int n_ranks, int rank,
                            kt_sharded **out);
int kt_sharded_connect_rccl(kt_sharded *s, const uint8_t *id128);
int kt_sharded_connect_host(kt_sharded *s, kt_alltoall_fn fn, void *user);
int kt_sharded_destroy(kt_sharded *s);
int kt_sharded_clear(kt_sharded *s);

/* collective.  replaces: count_chunk's loop for one chunk of this rank's records, counter/src/lib.rs:119-131 */
int kt_sharded_add_reads(kt_sharded *s, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int mem);

/* collective; call once after the last kt_sharded_add_reads and before reading the shard: delivers the k-mers that
 * did not fit their exchange region (batches dominated by few k-mers).  replaces: merge, counter/src/lib.rs:172-234 */
int kt_sharded_finalize(kt_sharded *s);

/* this rank's shard, an ordinary table: kt_ctr_size / kt_ctr_export / kt_cov_batch work on it (owned by `s`) */
int kt_sharded_table(kt_sharded *s, kt_ctr **table);

/* bytes this rank has sent to other ranks so far (a single rank made to route into n regions, KT_SHARD_FORCE = n: the
 * bytes of the regions a rank of n would have sent) */
int kt_sharded_exchanged_bytes(kt_sharded *s, uint64_t *bytes);

/* what the route pass of this rank's LAST batch put into every owner's region: records[o] records holding kmers[o] k-mers
 * (arrays of 64; *n_owners = the owners: n_ranks, or KT_SHARD_FORCE's regions of a single rank).  max / mean of kmers[]
 * over the owners is the imbalance the minimiser ownership costs.  (no reference counterpart: statistics) */
int kt_sharded_route_stats(kt_sharded *s, uint32_t *n_owners, uint64_t *records, uint64_t *kmers);

/* what carries the exchange: *n_ranks = the ranks this counter was created for; *rccl_ranks = what the library's own
 * RCCL communicator reports (ncclCommCount on the communicator kt_sharded_connect_rccl made; 0 when the transport is the
 * caller's host all-to-all or the counter has a single rank and no communicator); *transport = 0 none (one rank, no
 * exchange), 1 librccl ncclSend / ncclRecv, 2 the caller's host all-to-all.  A benchmark line that claims N GPUs can
 * show that RCCL saw N ranks.  (no reference counterpart: statistics) */
int kt_sharded_comm_info(kt_sharded *s, int *n_ranks, int *rccl_ranks, int *transport);

/* the rank that owns a canonical k-mer in this sharded table (host helper, the function the device uses) */
int kt_sharded_owner_of(kt_sharded *s, uint64_t kmer, uint32_t *owner);

/* The same without a counter or a GPU: the minimiser length m and the window w = k - m + 1 (m-mers per k-mer) the
 * sharded counter uses for k-mers of length k, and the rank that owns a k-mer (either strand) among n_ranks.
 * replaces: `min_mer % n_parts`, counter/src/lib.rs:127 */
int kt_shard_minimiser(int k, uint32_t *m, uint32_t *w);
uint32_t kt_shard_owner_of(uint64_t kmer, int k, uint32_t n_ranks);

/* hash partition of a canonical k-mer among n_owners (host helper, same function the device uses): what
 * kt_ctr_add_reads_part and kt_ctr_route split by - the LOW hash bits, independent of a table's slot bits. */
uint32_t kt_owner_of(uint64_t kmer, uint32_t n_owners);

/* Synthetic reads for benchmarks/parity, generated in HBM:
 * bases_dev[n_reads*read_len], offsets_dev[n_reads+1] (may be NULL).  Deterministic
 * in (seed, first_read + i, pos); genome_len == 0: i.i.d. uniform ACGT, else reads
 * sampled from a random genome_len-bp genome (both strands, 1 % substitutions);
 * noise: ~0.1 % N, ~1 % lower case.  Device pointers only. */
int kt_synth_reads(kt_ctx *ctx, uint64_t seed, uint64_t first_read, uint64_t n_reads,
                   uint32_t read_len, int noise, uint64_t genome_len, uint8_t *bases_dev,
                   uint64_t *offsets_dev);

#ifdef __cplusplus
}
#endif
#endif /* KMERTOOLS_HIP_H */
