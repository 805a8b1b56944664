#include "computers.hpp"

#include <stdio.h>
#include <string.h>

#include <charconv>
#include <thread>

#include "seqio.hpp"

namespace kthost {

void append_fixed6(std::string &out, double x) {
    char buf[64];
    const int n = snprintf(buf, sizeof buf, "%.6f", x);  // exact, correctly rounded like Rust's {:.6}
    out.append(buf, (size_t)n);
}

void append_display(std::string &out, double x) {
    char buf[400];
    const auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::fixed);  // shortest round-trip
    out.append(buf, (size_t)(r.ptr - buf));
}

Device::~Device() {
    if (ctx) kt_ctx_destroy(ctx);
}

std::string Device::ensure() {
    if (ctx) return "";
    if (kt_ctx_create(index, nullptr, 1, &ctx) != KT_OK) return std::string(kt_last_error());
    return "";
}

static int worker_count(int threads) {
    if (threads > 0) return threads;
    const unsigned h = std::thread::hardware_concurrency();
    return h ? (int)h : 1;
}

// format rows [0, n) with `fn(row, out)` on `threads` workers, return the pieces in row order
template <class F>
static void format_rows(uint64_t n, int threads, std::vector<std::string> &pieces, F fn) {
    int t = worker_count(threads);
    if ((uint64_t)t > n) t = n ? (int)n : 1;
    pieces.assign((size_t)t, std::string());
    std::vector<std::thread> pool;
    for (int w = 0; w < t; w++) {
        pool.emplace_back([&, w] {
            const uint64_t lo = n * (uint64_t)w / (uint64_t)t, hi = n * (uint64_t)(w + 1) / (uint64_t)t;
            std::string &s = pieces[(size_t)w];
            for (uint64_t r = lo; r < hi; r++) fn(r, s);
        });
    }
    for (auto &th : pool) th.join();
}

static uint64_t batch_bases(uint64_t memory) {
    const uint64_t cap = 256ull << 20;  // keeps the f64 output slab of a batch bounded
    return memory < cap ? (memory ? memory : 1) : cap;
}

// ---------------------------------------------------------------------------------------------
OligoComputer::OligoComputer(std::string in_path, std::string out_path, int ksize, bool count_min)
    : in_path_(std::move(in_path)), out_path_(std::move(out_path)), ksize_(ksize), count_min_(count_min) {}

std::vector<std::string> OligoComputer::get_header() const {
    std::vector<std::string> h;
    char buf[40];
    if (count_min_) {
        uint64_t bins = 0;
        kt_bins(ksize_, 1, &bins);
        std::vector<uint64_t> canon(bins);
        uint32_t kcount = 0;
        kt_pos_map(ksize_, nullptr, canon.data(), &kcount);
        for (uint64_t i = 0; i < kcount; i++) {
            kt_numeric_to_kmer(canon[i], ksize_, buf);
            h.emplace_back(buf);
        }
    } else {
        const uint64_t n = 1ull << (2 * ksize_);
        for (uint64_t km = 0; km < n; km++) {
            kt_numeric_to_kmer(km, ksize_, buf);
            h.emplace_back(buf);
        }
    }
    return h;
}

std::string OligoComputer::vectorise() {
    SeqReader reader;
    // the reference sniffs the first byte on its batch path (stdin or raw counts) and trusts the
    // extension on its mmap path; an unknown extension falls back to sniffing here
    if (!reader.open(in_path_, in_path_ == "-" || !norm_)) return reader.error();
    FILE *out = fopen(out_path_.c_str(), "wb");
    if (!out) return "Unable to write to file: " + out_path_;
    if (std::string e = dev_.ensure(); !e.empty()) {
        fclose(out);
        return e;
    }
    uint64_t bins = 0;
    kt_bins(ksize_, count_min_, &bins);
    if (header_) {
        std::string line;
        const auto h = get_header();
        for (size_t i = 0; i < h.size(); i++) {
            if (i) line += delim_;
            line += h[i];
        }
        line += "\n";
        fwrite(line.data(), 1, line.size(), out);
    }
    Batch b;
    std::vector<double> rows;
    std::vector<std::string> pieces;
    std::string err;
    for (;;) {
        const bool more = reader.next_batch(b, batch_bases(memory_), 1ull << 20);
        const uint64_t n = b.n_reads();
        if (n) {
            rows.resize(n * bins);
            if (kt_oligo_batch(dev_.ctx, b.bases.empty() ? (const uint8_t *)"" : b.bases.data(), b.offsets.data(), n,
                               ksize_, count_min_, norm_, 1, KT_F64, rows.data(), KT_MEM_HOST) != KT_OK) {
                err = kt_last_error();
                break;
            }
            const bool norm = norm_;
            const std::string &delim = delim_;
            format_rows(n, threads_, pieces, [&](uint64_t r, std::string &s) {
                const double *row = rows.data() + r * bins;
                for (uint64_t i = 0; i < bins; i++) {
                    if (i) s += delim;
                    if (norm) append_fixed6(s, row[i]); else append_display(s, row[i]);
                }
                s += '\n';
            });
            for (const auto &p : pieces) fwrite(p.data(), 1, p.size(), out);
        }
        if (!more) break;
    }
    if (err.empty() && reader.failed()) err = reader.error();
    fclose(out);
    return err;
}

// ---------------------------------------------------------------------------------------------
OligoCgrComputer::OligoCgrComputer(std::string in_path, std::string out_path, int ksize, uint64_t vecsize)
    : in_path_(std::move(in_path)), out_path_(std::move(out_path)), ksize_(ksize), vecsize_(vecsize) {}

std::string OligoCgrComputer::vectorise() {
    SeqReader reader;
    if (!reader.open(in_path_, true)) return reader.error();  // oligocgr.rs:64-72 sniffs '>'
    FILE *out = fopen(out_path_.c_str(), "wb");
    if (!out) return "Unable to write to file: " + out_path_;
    if (std::string e = dev_.ensure(); !e.empty()) {
        fclose(out);
        return e;
    }
    uint64_t bins = 0;
    kt_bins(ksize_, 1, &bins);
    // the CGR point of every canonical k-mer is read-independent: build "(x,y," once
    std::vector<double> xy(bins * 2);
    kt_cgr_coords(ksize_, (double)vecsize_, xy.data());
    std::vector<std::string> prefix(bins);
    for (uint64_t i = 0; i < bins; i++) {
        std::string &p = prefix[i];
        p = "(";
        append_display(p, xy[2 * i]);
        p += ",";
        append_display(p, xy[2 * i + 1]);
        p += ",";
    }
    Batch b;
    std::vector<double> rows;
    std::vector<std::string> pieces;
    std::string err;
    for (;;) {
        // rows are 8 * bins bytes each: bound the batch by reads as well
        const uint64_t max_reads = bins >= 2048 ? 8192 : 262144;
        const bool more = reader.next_batch(b, batch_bases(memory_), max_reads);
        const uint64_t n = b.n_reads();
        if (n) {
            rows.resize(n * bins);
            if (kt_oligo_batch(dev_.ctx, b.bases.empty() ? (const uint8_t *)"" : b.bases.data(), b.offsets.data(), n,
                               ksize_, 1, norm_, 1, KT_F64, rows.data(), KT_MEM_HOST) != KT_OK) {
                err = kt_last_error();
                break;
            }
            format_rows(n, threads_, pieces, [&](uint64_t r, std::string &s) {
                const double *row = rows.data() + r * bins;
                for (uint64_t i = 0; i < bins; i++) {
                    if (i) s += ' ';
                    s += prefix[i];
                    append_display(s, row[i]);
                    s += ')';
                }
                s += '\n';
            });
            for (const auto &p : pieces) fwrite(p.data(), 1, p.size(), out);
        }
        if (!more) break;
    }
    if (err.empty() && reader.failed()) err = reader.error();
    fclose(out);
    return err;
}

// ---------------------------------------------------------------------------------------------
CountComputer::CountComputer(std::string in_path, std::string out_dir, int ksize)
    : in_path_(std::move(in_path)), out_dir_(std::move(out_dir)), ksize_(ksize) {}

CountComputer::~CountComputer() {
    if (ctr_) kt_ctr_destroy(ctr_);
}

std::string CountComputer::count() {
    // init(): pre-pass for record count and total length (counter/src/lib.rs:236-249); here it
    // sizes the HBM table instead of the reference's partition count
    std::string err;
    if (!SeqReader::seq_stats(in_path_, seq_count_, total_length_, err)) return err;
    if (std::string e = dev_.ensure(); !e.empty()) return e;
    uint64_t max_distinct = total_length_;
    if (ksize_ <= 15) {
        const uint64_t n4k = 1ull << (2 * ksize_);
        const uint64_t canon = (ksize_ & 1) ? n4k / 2 : (n4k + (1ull << ksize_)) / 2;
        if (canon < max_distinct) max_distinct = canon;
    }
    uint64_t cap = 1024;
    while (cap < 2 * max_distinct) cap <<= 1;
    if (kt_ctr_create(dev_.ctx, ksize_, cap, &ctr_) != KT_OK) return kt_last_error();
    SeqReader reader;
    if (!reader.open(in_path_, false)) return reader.error();
    Batch b;
    for (;;) {
        const bool more = reader.next_batch(b, 256ull << 20, 1ull << 22);
        if (b.n_reads() && !b.bases.empty()) {
            if (kt_ctr_add_reads(ctr_, b.bases.data(), b.offsets.data(), b.n_reads(), KT_MEM_HOST) != KT_OK)
                return kt_last_error();
        }
        if (!more) break;
    }
    if (reader.failed()) return reader.error();
    return "";
}

std::string CountComputer::merge(bool /*del: no temp files exist to delete*/) {
    if (!ctr_) return "count() has not run";
    uint64_t n = 0;
    if (kt_ctr_size(ctr_, &n) != KT_OK) return kt_last_error();
    std::vector<uint64_t> keys(n ? n : 1);
    std::vector<uint32_t> counts(n ? n : 1);
    uint64_t got = 0;
    if (n && kt_ctr_export(ctr_, keys.data(), counts.data(), n, &got, KT_MEM_HOST) != KT_OK) return kt_last_error();
    const std::string path = out_dir_ + "/kmers.counts";
    FILE *out = fopen(path.c_str(), "wb");
    if (!out) return "Unable to write to file: " + path;
    std::vector<std::string> pieces;
    const bool acgt = acgt_;
    const int k = ksize_;
    format_rows(got, threads_, pieces, [&](uint64_t i, std::string &s) {
        char buf[40];
        if (acgt) {
            kt_numeric_to_kmer(keys[i], k, buf);  // counter/src/lib.rs:221-226
            s += buf;
        } else {
            const auto r = std::to_chars(buf, buf + sizeof buf, keys[i]);
            s.append(buf, (size_t)(r.ptr - buf));
        }
        s += '\t';
        const auto r2 = std::to_chars(buf, buf + sizeof buf, counts[i]);
        s.append(buf, (size_t)(r2.ptr - buf));
        s += '\n';
    });
    for (const auto &p : pieces) fwrite(p.data(), 1, p.size(), out);
    fclose(out);
    return "";
}

// ---------------------------------------------------------------------------------------------
CovComputer::CovComputer(std::string in_path, std::string out_dir, int ksize, uint64_t bin_size, uint64_t bin_count)
    : in_path_(in_path), in_path_kmer_(std::move(in_path)), out_dir_(std::move(out_dir)), ksize_(ksize),
      bin_size_(bin_size), bin_count_(bin_count) {}

CovComputer::~CovComputer() { delete ctr_; }

std::string CovComputer::build_table() {
    delete ctr_;
    ctr_ = new CountComputer(in_path_kmer_, out_dir_, ksize_);
    ctr_->set_threads(threads_);
    ctr_->set_max_memory(memory_ceil_gb_);
    ctr_->set_device(device_);
    std::string e = ctr_->count();
    if (e.empty()) e = ctr_->merge(true);  // the reference leaves kmers.counts behind as well
    return e;
}

std::string CovComputer::compute_coverages() {
    // the reference parses kmers.counts back into a HashMap (:82-92); the table is still in HBM here
    if (!ctr_ || !ctr_->table()) return "build_table() has not run";
    SeqReader reader;
    if (!reader.open(in_path_, false)) return reader.error();
    const std::string path = out_dir_ + "/kmers.vectors";
    FILE *out = fopen(path.c_str(), "wb");
    if (!out) return "Unable to write to file: " + path;
    Batch b;
    std::vector<double> rows;
    std::vector<std::string> pieces;
    std::string err;
    const uint64_t bins = bin_count_;
    for (;;) {
        const uint64_t max_reads = bins >= 2048 ? 8192 : 1ull << 20;
        const bool more = reader.next_batch(b, 256ull << 20, max_reads);
        const uint64_t n = b.n_reads();
        if (n) {
            rows.resize(n * bins);
            if (kt_cov_batch(ctr_->table(), b.bases.empty() ? (const uint8_t *)"" : b.bases.data(), b.offsets.data(), n,
                             bin_size_, bin_count_, norm_, KT_F64, rows.data(), KT_MEM_HOST) != KT_OK) {
                err = kt_last_error();
                break;
            }
            const bool norm = norm_;
            const std::string &delim = delim_;
            format_rows(n, threads_, pieces, [&](uint64_t r, std::string &s) {
                const double *row = rows.data() + r * bins;
                for (uint64_t i = 0; i < bins; i++) {
                    if (i) s += delim;
                    if (norm) append_fixed6(s, row[i]); else append_display(s, row[i]);  // :116-120
                }
                s += '\n';
            });
            for (const auto &p : pieces) fwrite(p.data(), 1, p.size(), out);
        }
        if (!more) break;
    }
    if (err.empty() && reader.failed()) err = reader.error();
    fclose(out);
    return err;
}

}  // namespace kthost
