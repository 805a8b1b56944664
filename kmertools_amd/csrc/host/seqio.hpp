// seqio.hpp - FASTA/FASTQ(/.gz/stdin) record reader producing CSR batches for the C ABI.
//
// Host-side counterpart of the reference's ktio::seq (ktio/src/seq.rs:12-155), which wraps
// rust-bio 2.3.0's fasta/fastq readers (third-party, Cargo.lock:98; not in the reference tree -
// behaviour restated from its documented format handling and pinned by the reference's reader
// tests, ktio/src/seq.rs:165-233):
//   * format: by file extension after stripping ".gz" (.fq/.fastq | .fasta/.fa/.fna,
//     SeqFormat::get, seq.rs:30-41), else by the first byte ('>' = FASTA; the reference's batch
//     paths sniff the same way, composition/src/oligo.rs:100-105)
//   * ".gz" suffix => gzip (get_reader, seq.rs:141-155); "-" => stdin
//   * FASTA: header '>' + id (first whitespace-delimited token); sequence lines are joined with
//     trailing whitespace removed; bytes are otherwise untouched (case preserved)
//   * FASTQ: '@' header, sequence lines up to the '+' line, then as many quality lines
// Unlike the reference (one heap Sequence per record, single consumer behind a mutex) records
// are appended straight into a reusable batch: `bases` (concatenated) + `offsets`.
// Plain (uncompressed, seekable) files of some size are mapped and parsed by several threads: the file is cut at
// record boundaries into pieces of ~32 MB, a pool parses pieces concurrently, and next_batch hands the pieces'
// batches out in file order (KT_READER_THREADS=1 forces the serial reader; gzip and stdin always take it).
#pragma once
#include <stdint.h>
#include <zlib.h>

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace kthost {

enum class SeqFormat { Fasta, Fastq, Unknown };

SeqFormat format_from_path(const std::string &path);  // SeqFormat::get

struct Batch {
    std::vector<uint8_t> bases;
    std::vector<uint64_t> offsets;  // n + 1 entries, offsets[0] == 0
    std::vector<std::string> ids;   // filled only when keep_ids
    uint64_t first_record = 0;      // ordinal of the batch's first record (Sequence::n)
    void clear() {
        bases.clear();
        offsets.assign(1, 0);
        ids.clear();
    }
    uint64_t n_reads() const { return offsets.empty() ? 0 : offsets.size() - 1; }
};

class SeqReader {
  public:
    SeqReader() = default;
    ~SeqReader();
    SeqReader(const SeqReader &) = delete;
    SeqReader &operator=(const SeqReader &) = delete;

    // Opens `path` ("-" = stdin).  sniff = decide the format from the first byte instead of the
    // extension.  Returns false and sets error() on failure ("Unable to open: {path}", seq.rs:147).
    bool open(const std::string &path, bool sniff);
    // Appends records until the batch holds >= max_bases bases or max_reads reads.
    // Returns false at end of input (the batch may still hold records) or on a parse error.
    bool next_batch(Batch &b, uint64_t max_bases, uint64_t max_reads, bool keep_ids = false);
    // Back to the first record, keeping threads and buffers (false: the input cannot be read twice - stdin)
    bool rewind();
    size_t buffer_bytes();  // what the reader's own buffers hold right now
    bool failed() const { return !err_.empty(); }
    const std::string &error() const { return err_; }
    uint64_t records_read() const { return n_records_; }

    // Sequences::seq_stats (seq.rs:69-94): record count and total length of a whole file
    static bool seq_stats(const std::string &path, uint64_t &seq_count, uint64_t &total_length, std::string &err);

    // parses the records that lie in [p, p + n) of a mapped file (the parallel reader's workers)
    void open_range(const unsigned char *p, size_t n, SeqFormat fmt);

  private:
    struct Parallel;  // the mapped file, its pieces, the worker pool (seqio.cpp)
    std::shared_ptr<Parallel> par_;
    bool next_batch_parallel(Batch &b, bool keep_ids);
    bool fill();
    bool read_line(std::string &line);  // without the trailing '\n'; false at EOF
    // same, without a copy when the whole line sits in the read buffer: [p, p + n) stays valid until the next call
    bool read_line_view(const char *&p, size_t &n);
    bool peek(int &c);
    gzFile gz_ = nullptr;  // zlib reads plain files transparently
    std::vector<unsigned char> buf_;
    const unsigned char *base_ = nullptr;  // what is being parsed: buf_ (stream) or a mapped range
    bool mem_ = false;                     // a fixed memory range: nothing to refill
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false;
    SeqFormat fmt_ = SeqFormat::Unknown;
    std::string err_, line_, pending_;
    bool have_pending_ = false;
    uint64_t n_records_ = 0;
};

}  // namespace kthost
