// computers.hpp - C++ mirror of the reference's Rust-level operator API for the hot path:
//   OligoComputer      composition/src/oligo.rs:15-93
//   OligoCgrComputer   composition/src/oligocgr.rs:16-121
//   CgrComputer        composition/src/cgr.rs:42-144
//   CountComputer      counter/src/lib.rs:22-90, 172-234
//   CovComputer        coverage/src/lib.rs:14-184
//   seq_to_min / bin_sequences   misc/src/minimisers.rs:11-160 (free functions there too)
// Same constructor arguments, setters and entry points (vectorise / count / merge); the
// per-read / per-k-mer work goes through the C ABI (include/kmertools_hip.h) to the GPU.
// The reference's error style is kept: vectorise() returns "" on success or the message that
// the CLI prints after "Error: ".
#pragma once
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "../../../include/kmertools_hip.h"

namespace kthost {

// Rust `format!("{:.6}", x)` (oligo.rs:132-134) and `Display` for f64 (shortest round-trip,
// positional notation, integral values without ".0": oligo.rs:136, oligocgr.rs:95)
constexpr size_t FIXED6_BUF = 352;  // "%.6f" of DBL_MAX is 316 characters
size_t format_fixed6(char *buf, double x);  // buf holds FIXED6_BUF bytes; returns the length written
void append_fixed6(std::string &out, double x);
void append_display(std::string &out, double x);

double debug_emit_bench(uint64_t n_rows, uint64_t bins, bool norm, int threads, int reps);

struct Device {  // one kt_ctx per computer, created on first use
    int index = 0;
    kt_ctx *ctx = nullptr;
    ~Device();
    std::string ensure();  // "" or error message
};

class OligoComputer {
  public:
    OligoComputer(std::string in_path, std::string out_path, int ksize, bool count_min);
    void set_threads(int t) { threads_ = t; }
    void set_norm(bool n) { norm_ = n; }
    void set_delim(std::string d) { delim_ = std::move(d); }
    void set_max_memory(uint64_t m) { memory_ = m; }
    void set_header(bool h) { header_ = h; }
    void set_device(int d) { dev_.index = d; }
    std::vector<std::string> get_header() const;  // oligo.rs:69-83
    std::string vectorise();                      // oligo.rs:88-93 (batch and mmap paths write identical bytes)

  private:
    std::string in_path_, out_path_, delim_ = " ";
    int ksize_, threads_ = 0;
    bool count_min_, norm_ = true, header_ = false;
    uint64_t memory_ = 4ull << 30;  // GB_4, oligo.rs:13
    Device dev_;
};

class OligoCgrComputer {
  public:
    OligoCgrComputer(std::string in_path, std::string out_path, int ksize, uint64_t vecsize);
    void set_threads(int t) { threads_ = t; }
    void set_norm(bool n) { norm_ = n; }
    void set_device(int d) { dev_.index = d; }
    std::string vectorise();  // oligocgr.rs:63-121

  private:
    std::string in_path_, out_path_;
    int ksize_, threads_ = 0;
    uint64_t vecsize_;
    bool norm_ = true;
    uint64_t memory_ = 4ull << 30;
    Device dev_;
};

class CgrComputer {
  public:
    CgrComputer(std::string in_path, std::string out_path, uint64_t vecsize);
    void set_threads(int t) { threads_ = t; }
    void set_device(int d) { dev_.index = d; }
    // cgr.rs:67-125.  "" on success; "Bad nucleotide, unable to proceed" where the reference's
    // worker unwraps that Err (:95) and the process dies
    std::string vectorise();

  private:
    std::string in_path_, out_path_;
    uint64_t vecsize_;
    int threads_ = 0;
    Device dev_;
};

class CountComputer {
  public:
    CountComputer(std::string in_path, std::string out_dir, int ksize);
    ~CountComputer();
    void set_threads(int t) { threads_ = t; }
    void set_max_memory(double gb) { memory_ceil_gb_ = gb; }  // the host ceiling: the table is written out in slabs sized by it
    void set_acgt_output(bool a) { acgt_ = a; }
    void set_device(int d) { dev_.index = d; }
    void set_devices(int n) { n_devices_ = n < 1 ? 1 : n; }  // --devices N: the table sharded over N GPUs (kt_sharded_*)
    // counter/src/lib.rs:69-90.  No temp files: one resident table - or, when the distinct k-mers cannot fit the HBM,
    // `passes()` passes over the input, one hash partition each, written to kmers.counts as they complete.
    std::string count();
    std::string merge(bool del);     // counter/src/lib.rs:172-234: writes {out_dir}/kmers.counts
    uint64_t seq_count() const { return seq_count_; }  // 0 when the sizing pre-pass was skipped (plain files)
    uint32_t passes() const { return passes_; }
    // out of core (passes() > 1): called with every pass's complete table before it is written out and cleared
    void set_pass_hook(std::function<std::string(uint32_t pass, uint32_t passes, kt_ctr *table)> h) { pass_hook_ = std::move(h); }
    kt_ctr *table() const { return passes_ == 1 && !sharded_done_ ? ctr_ : nullptr; }  // the resident table (after count())
    kt_ctx *context() const { return dev_.ctx; }
    // --devices N, for a caller that looks k-mers up afterwards (cov): merge() writes the shards but leaves them on their GPUs
    void set_keep_shards(bool k) { keep_shards_ = k; }
    size_t n_shards() const { return sharded_done_ ? shards_.size() : 0; }
    kt_ctr *shard_table(size_t r) const;  // shard r's table (an ordinary kt_ctr that answers for its own k-mers only)

  private:
    std::string in_path_, out_dir_;
    int ksize_, threads_ = 0;
    double memory_ceil_gb_ = 6.0;
    bool acgt_ = false;
    uint64_t seq_count_ = 0, total_length_ = 0;
    Device dev_;
    kt_ctr *ctr_ = nullptr;
    int n_devices_ = 1;
    uint32_t passes_ = 1;
    bool sharded_done_ = false, keep_shards_ = false;
    // --devices N: the shards stay on their GPUs after count(); merge() writes them out one after the other, in slabs
    std::vector<kt_sharded *> shards_;
    std::vector<kt_ctx *> shard_ctx_;
    void release_shards();
    std::function<std::string(uint32_t, uint32_t, kt_ctr *)> pass_hook_;
    std::string count_sharded(uint64_t max_distinct);
};

// `min -p s2m`: one line per record, "id\tMMER:start-end\t...\t\n" (misc/src/minimisers.rs:93-160).
// `min -p m2s`: one line per minimiser, "MMER\t[(\"id\", start, end), ...]\n" - Rust's {:?} of the
// Vec<(String, usize, usize)> (:11-91).  The reference's line order (and, for m2s, the order inside a
// vector) depends on thread timing; here lines follow the input (s2m) or ascending minimiser text (m2s).
// wsize 0 = one minimiser per sequence.  Return "" or the error message.
std::string seq_to_min(uint64_t wsize, int msize, const std::string &in_path, const std::string &out_path, int threads,
                       int device = 0);
std::string bin_sequences(uint64_t wsize, int msize, const std::string &in_path, const std::string &out_path, int threads,
                          int device = 0);

class CovComputer {
  public:
    CovComputer(std::string in_path, std::string out_dir, int ksize, uint64_t bin_size, uint64_t bin_count);
    void set_threads(int t) { threads_ = t; }
    void set_norm(bool n) { norm_ = n; }
    void set_delim(std::string d) { delim_ = std::move(d); }
    void set_kmer_path(std::string p) { in_path_kmer_ = std::move(p); }
    void set_max_memory(double gb) { memory_ceil_gb_ = gb; }
    void set_device(int d) { device_ = d; }
    void set_devices(int n) { n_devices_ = n < 1 ? 1 : n; }  // --devices N: the table sharded over N GPUs, every read looked up in every shard
    std::string build_table();        // coverage/src/lib.rs:69-77: count + merge -> {out_dir}/kmers.counts
    std::string compute_coverages();  // coverage/src/lib.rs:79-163: writes {out_dir}/kmers.vectors

  private:
    std::string in_path_, in_path_kmer_, out_dir_, delim_ = " ";
    int ksize_, threads_ = 0, device_ = 0, n_devices_ = 1;
    uint64_t bin_size_, bin_count_;
    bool norm_ = true;
    double memory_ceil_gb_ = 6.0;
    CountComputer *ctr_ = nullptr;  // owns the HBM table the coverages are looked up in
    // a table that needed several passes (the k-mers do not fit the HBM): the reads' raw bin counts, summed over the
    // passes as each pass's table is complete (kt_cov_batch_part), normalised and written at the end
    std::vector<uint32_t> acc_rows_;
    uint64_t acc_reads_ = 0;
    std::string cov_pass(uint32_t pass, uint32_t passes, kt_ctr *table);

  public:
    ~CovComputer();
    CovComputer(const CovComputer &) = delete;
    CovComputer &operator=(const CovComputer &) = delete;
};

}  // namespace kthost
