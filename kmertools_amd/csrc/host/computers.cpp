#include "computers.hpp"

#include <malloc.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

#include "seqio.hpp"

namespace kthost {

// ---- text formats ------------------------------------------------------------------------------
// Rust's `{:.6}` rounds the EXACT binary value half-to-even (core::num::flt2dec dragon
// format_exact), which is also what glibc's "%.6f" does; snprintf costs ~150 ns and serialises
// threads on locale state, so the common range is done by hand: N = round_half_even(x * 10^6)
// exactly, from the rounded product p and its exact error e (Dekker two-product; 10^6 has 20
// significant bits so only x needs splitting), then "<N / 10^6>.<6 digits>".
static const char DIGIT_PAIRS[] =
    "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
    "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";

size_t format_fixed6(char *buf, double x) {
    if (!(x >= 0.0) || x >= 4.0e9) return (size_t)snprintf(buf, FIXED6_BUF, "%.6f", x);  // negative, NaN, inf, huge
    const double p = x * 1e6;
    const double c = 134217729.0 * x;  // 2^27 + 1
    const double xh = c - (c - x), xl = x - xh;
    const double e = (xh * 1e6 - p) + xl * 1e6;  // x * 10^6 == p + e exactly
    const double n = floor(p);
    const double d = (p - n) - 0.5;  // exact: p < 2^52, so p - n and 0.5 are multiples of ulp(p); |e| <= ulp(p)/2
    uint64_t N = (uint64_t)n;
    if (d > 0.0 || (d == 0.0 && (e > 0.0 || (e == 0.0 && (N & 1ull))))) N += 1;
    const uint64_t ip = N / 1000000ull;
    const uint32_t fp = (uint32_t)(N % 1000000ull);
    char *q = buf;
    if (ip < 10) {
        *q++ = (char)('0' + ip);
    } else {
        q = std::to_chars(q, q + 24, ip).ptr;
    }
    *q++ = '.';
    memcpy(q, DIGIT_PAIRS + 2 * (fp / 10000u), 2);
    memcpy(q + 2, DIGIT_PAIRS + 2 * ((fp / 100u) % 100u), 2);
    memcpy(q + 4, DIGIT_PAIRS + 2 * (fp % 100u), 2);
    return (size_t)(q + 6 - buf);
}

void append_fixed6(std::string &out, double x) {
    char buf[FIXED6_BUF];
    out.append(buf, format_fixed6(buf, x));
}

void append_display(std::string &out, double x) {
    char buf[400];
    if (x >= 0.0 && x < 9.0e15 && !signbit(x) && x == (double)(uint64_t)x) {  // counts: integral values print bare
        const auto r = std::to_chars(buf, buf + sizeof buf, (uint64_t)x);
        out.append(buf, (size_t)(r.ptr - buf));
        return;
    }
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

// KT_CLI_TIMING=1: busy seconds of each stage on stderr (the stages overlap, see run_pipeline)
struct PhaseTimer {
    const char *what;
    double t[4] = {0, 0, 0, 0};  // read, device, format, write - each owned by one thread
    bool on;
    explicit PhaseTimer(const char *w) : what(w), on(getenv("KT_CLI_TIMING") != nullptr) {}
    ~PhaseTimer() {
        if (on) fprintf(stderr, "[timing] %s: read %.3f s, device %.3f s, format %.3f s, write %.3f s\n", what, t[0], t[1], t[2], t[3]);
    }
};
struct Lap {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double operator()() {
        const auto now = std::chrono::steady_clock::now();
        const double d = std::chrono::duration<double>(now - t0).count();
        t0 = now;
        return d;
    }
};

// CPUs this process may actually use: the hardware count capped by the cgroup CPU quota (a
// container with a 16-CPU quota on a 256-thread host is throttled, not sped up, by 256 workers)
static int effective_cpus() {
    static const int cached = [] {
        unsigned h = std::thread::hardware_concurrency();
        long n = h ? (long)h : 1;
        long quota = -1, period = -1;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
            char q[32];
            if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atol(q);
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
            if (fscanf(g, "%ld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *pf = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(pf, "%ld", &period) != 1) period = -1;
                fclose(pf);
            }
        }
        if (quota > 0 && period > 0) {
            const long c = (quota + period - 1) / period;
            if (c < n) n = c;
        }
        return (int)(n < 1 ? 1 : n);
    }();
    return cached;
}

static int worker_count(int threads) {
    if (threads > 0) return threads;  // -t: the caller's explicit choice
    return effective_cpus();
}

// format rows [0, n) with `fn(row, out)` on `threads` workers, return the pieces in row order
template <class F>
static void format_rows(uint64_t n, int threads, size_t bytes_per_row, std::vector<std::string> &pieces, F fn) {
    int t = worker_count(threads);
    const uint64_t min_rows = 2048;  // a thread is not worth starting for less
    if ((uint64_t)t > (n + min_rows - 1) / min_rows) t = (int)((n + min_rows - 1) / min_rows);
    if (t < 1) t = 1;
    // (never fewer pieces than before: a short last batch must not free the text buffers the next full one needs)
    if (pieces.size() < (size_t)t) pieces.resize((size_t)t);
    for (size_t w = (size_t)t; w < pieces.size(); w++) pieces[w].clear();
    auto work = [&](int w) {
        const uint64_t lo = n * (uint64_t)w / (uint64_t)t, hi = n * (uint64_t)(w + 1) / (uint64_t)t;
        // build in a thread-local object: the string headers in `pieces` share cache lines, and every
        // append updates its header (measured: 8 threads slower than 1 when appending in place)
        std::string s;
        s.swap(pieces[(size_t)w]);  // keeps the capacity of the previous batch
        s.clear();
        s.reserve((hi - lo) * bytes_per_row);
        for (uint64_t r = lo; r < hi; r++) fn(r, s);
        s.swap(pieces[(size_t)w]);
    };
    // every share runs on a spawned thread: a share run on the calling thread was measured 6x slower than
    // its siblings on the GPU box (new threads start on the caller's CPU and crowd it until they migrate)
    std::vector<std::thread> pool;
    for (int w = 0; w < t; w++) pool.emplace_back(work, w);
    for (auto &th : pool) th.join();
}

// ---- read -> device -> text pipeline ---------------------------------------------------------
// The three stages of every vectoriser (parse a batch of records; one C-ABI call that stages the
// batch to the GPU and brings the rows back; format + write the rows) run on their own threads
// over three rotating work items, so the file costs max(stage) instead of sum(stage).  Output
// order is the input order (one emit thread, FIFO hand-off).
struct Work {
    Batch b;
    std::vector<double> rows;
    std::vector<uint64_t> u64[4];  // minimisers: event offsets, m-mers, starts, ends
    uint64_t n_events = 0;

    // From its second batch on (i.e. for files of many batches, where the cost of locking is repaid) the row
    // buffer of a work item is page-locked, so the device-to-host copy of the rows - by far the largest
    // transfer - runs by DMA instead of through the driver's pageable staging.
    // (a resize that reallocates frees the registered buffer: release the page lock first, not afterwards)
    void resize_rows(size_t n) {
        if (n > rows.capacity()) unpin();
        rows.resize(n);
    }
    void pin_rows(kt_ctx *ctx) {
        if (++uses_ < 2) return;
        void *p = rows.data();
        const size_t bytes = rows.capacity() * sizeof(double);
        if (p == pin_ptr_ && bytes == pin_bytes_) return;
        unpin();
        if (bytes && kt_host_register(ctx, p, bytes) == KT_OK) {
            pin_ctx_ = ctx;
            pin_ptr_ = p;
            pin_bytes_ = bytes;
        }
    }
    void unpin() {
        if (pin_ptr_) kt_host_unregister(pin_ctx_, pin_ptr_);
        pin_ptr_ = nullptr;
        pin_bytes_ = 0;
    }
    // KT_CLI_PIN_BASES=1 (an experiment, off by default - DESIGN.md 6.1 has what it measured): the batch's bases buffer
    // page-locked as well, from the item's second batch on.  The parallel reader copies every batch into this buffer,
    // so its address is stable as long as no batch outgrows its capacity (reserved with slack at the first use).
    void pin_bases(kt_ctx *ctx) {
        static const bool on = getenv("KT_CLI_PIN_BASES") != nullptr;
        if (!on || uses_ < 2) return;
        void *p = b.bases.data();
        const size_t bytes = b.bases.capacity();
        if (p == bpin_ptr_ && bytes == bpin_bytes_) return;
        if (bpin_ptr_) kt_host_unregister(bpin_ctx_, bpin_ptr_);
        bpin_ptr_ = nullptr;
        if (bytes && kt_host_register(ctx, p, bytes) == KT_OK) {
            bpin_ctx_ = ctx;
            bpin_ptr_ = p;
            bpin_bytes_ = bytes;
        }
    }
    ~Work() {
        unpin();
        if (bpin_ptr_) kt_host_unregister(bpin_ctx_, bpin_ptr_);
    }

  private:
    uint64_t uses_ = 0;
    kt_ctx *pin_ctx_ = nullptr, *bpin_ctx_ = nullptr;
    void *pin_ptr_ = nullptr, *bpin_ptr_ = nullptr;
    size_t pin_bytes_ = 0, bpin_bytes_ = 0;
};

template <class T>
class Channel {
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<T> q_;
    bool closed_ = false;

  public:
    void push(T v) {
        {
            std::lock_guard<std::mutex> l(m_);
            q_.push_back(v);
        }
        cv_.notify_one();
    }
    void close() {
        {
            std::lock_guard<std::mutex> l(m_);
            closed_ = true;
        }
        cv_.notify_all();
    }
    bool pop(T &v) {  // false once closed and drained
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        v = q_.front();
        q_.pop_front();
        return true;
    }
};

// Writes formatted batches on its own thread, so formatting batch i + 1 overlaps the write of batch i.
// write() swaps the caller's pieces with a drained spare (capacities survive, nothing is reallocated)
// and blocks only when two batches are already waiting.
class AsyncWriter {
    using Pieces = std::vector<std::string>;
    FILE *out_;
    PhaseTimer &pt_;
    Pieces bufs_[3];
    Channel<Pieces *> todo_, spare_;
    std::thread th_;
    bool finished_ = false;

  public:
    AsyncWriter(FILE *out, PhaseTimer &pt) : out_(out), pt_(pt) {
        for (auto &b : bufs_) spare_.push(&b);
        th_ = std::thread([this] {
            Pieces *p = nullptr;
            while (todo_.pop(p)) {
                Lap lap;
                for (const auto &s : *p) fwrite(s.data(), 1, s.size(), out_);
                pt_.t[3] += lap();
                spare_.push(p);
            }
        });
    }
    void write(Pieces &pieces) {
        Pieces *p = nullptr;
        spare_.pop(p);
        p->swap(pieces);
        todo_.push(p);
    }
    void finish() {  // call before fclose(out)
        if (finished_) return;
        finished_ = true;
        todo_.close();
        th_.join();
    }
    ~AsyncWriter() { finish(); }
};

// Batch limits can be lowered from the environment (KT_CLI_BATCH_READS / KT_CLI_BATCH_BASES): the tests use it to
// push a small file through many batches (the rotating work items, the ordered writer, several table adds).
static uint64_t env_cap(const char *name, uint64_t v) {
    const char *e = getenv(name);
    if (!e || !*e) return v;
    const uint64_t c = strtoull(e, nullptr, 10);
    return c && c < v ? c : v;
}
static uint64_t cli_batch_reads(uint64_t v) { return env_cap("KT_CLI_BATCH_READS", v); }
static uint64_t cli_batch_bases(uint64_t v) { return env_cap("KT_CLI_BATCH_BASES", v); }

// device(w): fills w.rows from w.b, returns "" or an error message.  emit(w): writes the text.
static std::string run_pipeline(SeqReader &reader, uint64_t max_bases, uint64_t max_reads, PhaseTimer &pt,
                                const std::function<std::string(Work &)> &device,
                                const std::function<void(Work &)> &emit, bool keep_ids = false) {
    max_bases = cli_batch_bases(max_bases);
    max_reads = cli_batch_reads(max_reads);
    constexpr int DEPTH = 3;
    Work items[DEPTH];
    Channel<Work *> free_q, read_q, done_q;
    for (auto &w : items) free_q.push(&w);
    std::atomic<bool> stop{false};
    std::thread reader_thread([&] {
        Work *w = nullptr;
        bool more = true;
        Lap lap;
        while (more && !stop.load() && free_q.pop(w)) {
            lap();
            more = reader.next_batch(w->b, max_bases, max_reads, keep_ids);
            pt.t[0] += lap();
            if (w->b.n_reads()) read_q.push(w); else free_q.push(w);
        }
        read_q.close();
    });
    std::thread emit_thread([&] {
        Work *w = nullptr;
        while (done_q.pop(w)) {
            emit(*w);
            free_q.push(w);
        }
    });
    std::string err;
    Work *w = nullptr;
    while (read_q.pop(w)) {
        if (err.empty()) {
            Lap lap;
            err = device(*w);
            pt.t[1] += lap();
        }
        if (!err.empty()) {  // drain: let the reader see `stop` and finish
            stop.store(true);
            free_q.push(w);
            continue;
        }
        done_q.push(w);
    }
    done_q.close();
    reader_thread.join();
    emit_thread.join();
    if (err.empty() && reader.failed()) err = reader.error();
    return err;
}

static uint64_t batch_bases(uint64_t memory) {
    const uint64_t cap = 64ull << 20;  // small enough that a file of a few batches still overlaps its stages
    return memory < cap ? (memory ? memory : 1) : cap;
}

static const uint8_t *bases_ptr(const Batch &b) { return b.bases.empty() ? (const uint8_t *)"" : b.bases.data(); }

// writes `rows` (n x bins) as delimited text: {:.6} when normalised, Display otherwise
static void emit_matrix(AsyncWriter &writer, const Work &w, uint64_t bins, bool norm, const std::string &delim, int threads,
                        std::vector<std::string> &pieces, PhaseTimer &pt) {
    Lap lap;
    const double *rows = w.rows.data();
    format_rows(w.b.n_reads(), threads, bins * (norm ? 9 : 4) + 1, pieces, [&](uint64_t r, std::string &s) {
        const double *row = rows + r * bins;
        char buf[FIXED6_BUF];
        for (uint64_t i = 0; i < bins; i++) {
            if (i) s += delim;
            if (norm) s.append(buf, format_fixed6(buf, row[i])); else append_display(s, row[i]);
        }
        s += '\n';
    });
    pt.t[2] += lap();
    writer.write(pieces);
}

// the same for raw counts delivered as u32 (`-c`: half the device-to-host bytes of f64 rows, integers print bare)
static void emit_counts(AsyncWriter &writer, const Work &w, uint64_t bins, const std::string &delim, int threads,
                        std::vector<std::string> &pieces, PhaseTimer &pt) {
    Lap lap;
    const uint32_t *rows = reinterpret_cast<const uint32_t *>(w.rows.data());
    const char d0 = delim.size() == 1 ? delim[0] : ' ';
    format_rows(w.b.n_reads(), threads, bins * 4 + 1, pieces, [&](uint64_t r, std::string &s) {
        const uint32_t *row = rows + r * bins;
        char buf[16];
        for (uint64_t i = 0; i < bins; i++) {
            if (i) {
                if (delim.size() == 1) s += d0; else s += delim;
            }
            const uint32_t v = row[i];
            if (v < 10) {
                s += (char)('0' + v);
            } else {
                const auto res = std::to_chars(buf, buf + sizeof buf, v);
                s.append(buf, (size_t)(res.ptr - buf));
            }
        }
        s += '\n';
    });
    pt.t[2] += lap();
    writer.write(pieces);
}

// hidden `kmertools debug-emit`: times the text stage alone on fabricated count-ratio rows
double debug_emit_bench(uint64_t n_rows, uint64_t bins, bool norm, int threads, int reps) {
    Work w;
    w.b.clear();
    for (uint64_t r = 0; r < n_rows; r++) w.b.offsets.push_back(r + 1);
    w.resize_rows(n_rows * bins);
    for (uint64_t i = 0; i < n_rows * bins; i++) {
        const double c = (double)((i * 2654435761ull >> 7) % 5);
        w.rows[i] = norm ? c / 147.0 : c;
    }
    FILE *out = fopen("/dev/null", "wb");
    std::vector<std::string> pieces;
    PhaseTimer pt("debug-emit");
    double best = 1e30;
    for (int i = 0; i < reps; i++) {
        Lap lap;
        {
            AsyncWriter writer(out, pt);
            emit_matrix(writer, w, bins, norm, " ", threads, pieces, pt);
        }
        const double d = lap();
        if (d < best) best = d;
    }
    fclose(out);
    return best;
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
    // (the device context is created by the first device call: HIP start-up, 0.1-0.3 s, then runs while the reader
    // thread parses the first batch)
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
    std::vector<std::string> pieces;
    PhaseTimer pt("comp oligo");
    AsyncWriter writer(out, pt);
    const std::string err = run_pipeline(
        reader, batch_bases(memory_), 1ull << 19, pt,
        [&](Work &w) -> std::string {
            if (std::string e = dev_.ensure(); !e.empty()) return e;
            const uint64_t n = w.b.n_reads();
            // normalised rows are f64 (the reference's type, bit-identical); raw counts travel as u32
            w.resize_rows(norm_ ? n * bins : (n * bins + 1) / 2);
            w.pin_rows(dev_.ctx);
            w.pin_bases(dev_.ctx);
            if (kt_oligo_batch(dev_.ctx, bases_ptr(w.b), w.b.offsets.data(), n, ksize_, count_min_, norm_, 1,
                               norm_ ? KT_F64 : KT_U32, w.rows.data(), KT_MEM_HOST) != KT_OK)
                return kt_last_error();
            return "";
        },
        [&](Work &w) {
            if (norm_) emit_matrix(writer, w, bins, true, delim_, threads_, pieces, pt);
            else emit_counts(writer, w, bins, delim_, threads_, pieces, pt);
        });
    writer.finish();
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
    std::vector<std::string> pieces;
    PhaseTimer pt("comp cgr -k");
    AsyncWriter writer(out, pt);
    // rows are 8 * bins bytes each: bound the batch by reads as well
    const uint64_t max_reads = bins >= 2048 ? 8192 : 262144;
    const std::string err = run_pipeline(
        reader, batch_bases(memory_), max_reads, pt,
        [&](Work &w) -> std::string {
            const uint64_t n = w.b.n_reads();
            w.resize_rows(n * bins);
            w.pin_rows(dev_.ctx);
            w.pin_bases(dev_.ctx);
            if (kt_oligo_batch(dev_.ctx, bases_ptr(w.b), w.b.offsets.data(), n, ksize_, 1, norm_, 1, KT_F64, w.rows.data(),
                               KT_MEM_HOST) != KT_OK)
                return kt_last_error();
            return "";
        },
        [&](Work &w) {
            Lap lap;
            const double *rows = w.rows.data();
            format_rows(w.b.n_reads(), threads_, bins * 24, pieces, [&](uint64_t r, std::string &s) {
                const double *row = rows + r * bins;
                for (uint64_t i = 0; i < bins; i++) {
                    if (i) s += ' ';
                    s += prefix[i];
                    append_display(s, row[i]);
                    s += ')';
                }
                s += '\n';
            });
            pt.t[2] += lap();
            writer.write(pieces);
        });
    writer.finish();
    fclose(out);
    return err;
}

// ---------------------------------------------------------------------------------------------
CgrComputer::CgrComputer(std::string in_path, std::string out_path, uint64_t vecsize)
    : in_path_(std::move(in_path)), out_path_(std::move(out_path)), vecsize_(vecsize) {}

std::string CgrComputer::vectorise() {
    SeqReader reader;
    if (!reader.open(in_path_, true)) return reader.error();  // cgr.rs:68-76 sniffs '>'
    FILE *out = fopen(out_path_.c_str(), "wb");
    if (!out) return "Unable to write to file: " + out_path_;
    if (std::string e = dev_.ensure(); !e.empty()) {
        fclose(out);
        return e;
    }
    std::vector<std::string> pieces;
    PhaseTimer pt("comp cgr (whole sequence)");
    AsyncWriter writer(out, pt);
    // 16 bytes of points and ~40 bytes of text per base: keep the batches small
    const std::string err = run_pipeline(
        reader, 16ull << 20, 1ull << 18, pt,
        [&](Work &w) -> std::string {
            const uint64_t n = w.b.n_reads();
            w.resize_rows(2 * w.b.bases.size() + 2);
            w.pin_rows(dev_.ctx);
            w.pin_bases(dev_.ctx);
            if (kt_cgr_points(dev_.ctx, bases_ptr(w.b), w.b.offsets.data(), n, (double)vecsize_, w.rows.data(), nullptr,
                              KT_MEM_HOST) != KT_OK)
                return kt_last_error();
            return "";
        },
        [&](Work &w) {
            Lap lap;
            const double *xy = w.rows.data();
            const uint64_t *off = w.b.offsets.data();
            const uint64_t n = w.b.n_reads();
            const size_t per_row = n ? (size_t)(w.b.bases.size() / n + 1) * 44 : 1;
            format_rows(n, threads_, per_row, pieces, [&](uint64_t r, std::string &s) {
                for (uint64_t g = off[r]; g < off[r + 1]; g++) {  // "({},{})" joined by " " (:97-102)
                    if (g != off[r]) s += ' ';
                    s += '(';
                    append_display(s, xy[2 * g]);
                    s += ',';
                    append_display(s, xy[2 * g + 1]);
                    s += ')';
                }
                s += '\n';
            });
            pt.t[2] += lap();
            writer.write(pieces);
        });
    writer.finish();
    fclose(out);
    return err;
}

// ---------------------------------------------------------------------------------------------
CountComputer::CountComputer(std::string in_path, std::string out_dir, int ksize)
    : in_path_(std::move(in_path)), out_dir_(std::move(out_dir)), ksize_(ksize) {}

CountComputer::~CountComputer() {
    release_shards();
    if (ctr_) kt_ctr_destroy(ctr_);
}

kt_ctr *CountComputer::shard_table(size_t r) const {
    kt_ctr *t = nullptr;
    if (!sharded_done_ || r >= shards_.size() || !shards_[r] || kt_sharded_table(shards_[r], &t) != KT_OK) return nullptr;
    return t;
}

void CountComputer::release_shards() {
    for (auto *&sh : shards_)
        if (sh) kt_sharded_destroy(sh), sh = nullptr;
    for (auto *&c : shard_ctx_)
        if (c) kt_ctx_destroy(c), c = nullptr;
}

// What the process holds, from the kernel's and the allocator's books (KT_CLI_TIMING): VmHWM is the peak of the resident
// set - anonymous + file + shared pages, the HIP runtime's mappings among them -, `heap` is what malloc has handed out and
// not got back (mallinfo2: uordblks of the arenas + hblkhd of its own mmaps), which is what this program's buffers are.
struct MemReport {
    uint64_t hwm_kb = 0, rss_kb = 0, anon_kb = 0, file_kb = 0, shmem_kb = 0, heap_kb = 0, heap_free_kb = 0;
};
static uint64_t heap_in_use_kb() {
    const struct mallinfo2 mi = mallinfo2();
    return (uint64_t)(mi.uordblks + mi.hblkhd) >> 10;
}
static MemReport mem_report() {
    MemReport r;
    if (FILE *f = fopen("/proc/self/status", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f)) {
            if (strncmp(line, "VmHWM:", 6) == 0) r.hwm_kb = strtoull(line + 6, nullptr, 10);
            else if (strncmp(line, "VmRSS:", 6) == 0) r.rss_kb = strtoull(line + 6, nullptr, 10);
            else if (strncmp(line, "RssAnon:", 8) == 0) r.anon_kb = strtoull(line + 8, nullptr, 10);
            else if (strncmp(line, "RssFile:", 8) == 0) r.file_kb = strtoull(line + 8, nullptr, 10);
            else if (strncmp(line, "RssShmem:", 9) == 0) r.shmem_kb = strtoull(line + 9, nullptr, 10);
        }
        fclose(f);
    }
    const struct mallinfo2 mi = mallinfo2();
    r.heap_kb = (uint64_t)(mi.uordblks + mi.hblkhd) >> 10;
    r.heap_free_kb = (uint64_t)mi.fordblks >> 10;
    return r;
}
static uint64_t peak_rss_kb() { return mem_report().hwm_kb; }

// The table's lines, a slab at a time: the entries are staged on the device and fetched `slab` entries at a time, so the
// host holds one slab of (key, count) pairs and its text whatever the table's size - the reference's map.scan streams
// into the file the same way (counter/src/lib.rs:220-230).  `ceil_gb` (-m): the slab shrinks with the ceiling.
// One writer serves every slab of every pass (and every shard of --devices N): the pair arrays and the text pieces are
// allocated once and keep their capacity, so what a pass costs in host memory does not depend on how many passes went
// before it or on how many entries the pass holds (round 4 built them anew per slab: the allocator's retention of the
// freed blocks moved the process's peak by +-100 MB from run to run).
class TableWriter {
  public:
    TableWriter(bool acgt, int k, int threads, double ceil_gb) : acgt_(acgt), k_(k), threads_(threads) {
        // ~64 bytes of host memory per entry of a slab (12 of pairs, the rest text in flight): a million entries at a
        // time - 64 MB, far below any ceiling -m accepts (6 .. 128 GB); a sixteenth of the ceiling if that were smaller
        slab_ = (uint64_t)(ceil_gb * (double)(1ull << 30) / 16.0 / 64.0);
        if (slab_ > (1ull << 20)) slab_ = 1ull << 20;
        if (slab_ < (1ull << 16)) slab_ = 1ull << 16;
    }
    std::string write(FILE *out, kt_ctr *ctr, uint64_t *n_out) {
        uint64_t n = 0;
        if (kt_ctr_export_stage(ctr, &n) != KT_OK) return kt_last_error();
        if (n && keys_.empty()) {
            keys_.resize((size_t)slab_);
            counts_.resize((size_t)slab_);
        }
        for (uint64_t i0 = 0; i0 < n; i0 += slab_) {
            const uint64_t m = n - i0 < slab_ ? n - i0 : slab_;
            if (kt_ctr_export_fetch(ctr, i0, m, keys_.data(), counts_.data()) != KT_OK) return kt_last_error();
            // "{kmer}\t{count}\n" (or the ACGT form), counter/src/lib.rs:220-230
            const uint64_t *keys = keys_.data();
            const uint32_t *counts = counts_.data();
            format_rows(m, threads_, acgt_ ? (size_t)k_ + 13 : 32, pieces_, [&](uint64_t i, std::string &s) {
                char buf[40];
                if (acgt_) {
                    kt_numeric_to_kmer(keys[i], k_, buf);  // counter/src/lib.rs:221-226
                    s += buf;
                } else {
                    const auto rr = std::to_chars(buf, buf + sizeof buf, keys[i]);
                    s.append(buf, (size_t)(rr.ptr - buf));
                }
                s += '\t';
                const auto r2 = std::to_chars(buf, buf + sizeof buf, counts[i]);
                s.append(buf, (size_t)(r2.ptr - buf));
                s += '\n';
            });
            const uint64_t h = heap_in_use_kb();  // with a slab's pairs and its text in hand: the writer's high point
            if (h > heap_peak_kb_) heap_peak_kb_ = h;
            for (const auto &p : pieces_)
                if (fwrite(p.data(), 1, p.size(), out) != p.size()) return "Unable to write the table's lines";
            slabs_++;
        }
        entries_ += n;
        if (n_out) *n_out = n;
        return "";
    }
    // bytes held in the writer's own buffers (capacities: what stays allocated between slabs)
    uint64_t buffer_bytes() const {
        uint64_t b = keys_.capacity() * 8 + counts_.capacity() * 4;
        for (const auto &p : pieces_) b += p.capacity();
        return b;
    }
    uint64_t heap_peak_kb() const { return heap_peak_kb_; }
    uint64_t slab() const { return slab_; }
    uint64_t slabs() const { return slabs_; }

  private:
    bool acgt_;
    int k_, threads_;
    uint64_t slab_ = 0, slabs_ = 0, entries_ = 0, heap_peak_kb_ = 0;
    std::vector<uint64_t> keys_;
    std::vector<uint32_t> counts_;
    std::vector<std::string> pieces_;
};

static uint64_t env_u64_host(const char *name, uint64_t dflt) {
    const char *v = getenv(name);
    return v && *v ? strtoull(v, nullptr, 10) : dflt;
}

std::string CountComputer::count() {
    // init(): pre-pass for record count and total length (counter/src/lib.rs:236-249); here it
    // sizes the HBM table instead of the reference's partition count
    std::string err;
    Lap setup;
    if (in_path_ == "-") return "ctr reads its input more than once and cannot take stdin";  // (the reference panics
                                                                                            // on SeqFormat::get("-"))
    // An upper bound of the number of bases is all the sizing needs.  A plain file gives one without being read:
    // its size (FASTA), or half of it (FASTQ: as many quality bytes as bases).  Compressed input keeps the pre-pass.
    struct stat st;
    const bool plain = !(in_path_.size() > 3 && in_path_.compare(in_path_.size() - 3, 3, ".gz") == 0) &&
                       stat(in_path_.c_str(), &st) == 0 && S_ISREG(st.st_mode);
    if (plain) {
        seq_count_ = 0;
        total_length_ = format_from_path(in_path_) == SeqFormat::Fastq ? (uint64_t)st.st_size / 2 : (uint64_t)st.st_size;
    } else if (!SeqReader::seq_stats(in_path_, seq_count_, total_length_, err)) {
        return err;
    }
    const double t_stats = setup();
    uint64_t max_distinct = total_length_;
    if (ksize_ <= 15) {
        const uint64_t n4k = 1ull << (2 * ksize_);
        const uint64_t canon = (ksize_ & 1) ? n4k / 2 : (n4k + (1ull << ksize_)) / 2;
        if (canon < max_distinct) max_distinct = canon;
    }
    if (n_devices_ > 1) return count_sharded(max_distinct);
    if (std::string e = dev_.ensure(); !e.empty()) return e;
    const double t_dev = setup();
    // 1.9 slots per possible key = load factor ~0.5 at worst (the library rounds up to m * 2^j, m in 5..8)
    uint64_t want = max_distinct + max_distinct / 10 * 9;
    if (want < 1024) want = 1024;
    // The reference bounds its memory with -m: chunks of the input, partitions spilled to disk, merged partition by
    // partition (counter/src/lib.rs:114-118, 151-167, 188-231).  Here the bound is the HBM next to the build's
    // buffers; a table that cannot hold every distinct k-mer is filled in `passes` passes over the input, pass p
    // counting hash partition p only (kt_ctr_add_reads_part), each pass's table written out and cleared.
    const uint64_t batch_bases = 256ull << 20;
    uint64_t free_b = 0, total_b = 0, fit = want;
    if (kt_device_memory(dev_.ctx, &free_b, &total_b) == KT_OK) {
        const uint64_t reserve = batch_bases * 20 + (1ull << 30);  // two key arrays + staging of a batch
        const uint64_t usable = free_b > 2 * reserve ? free_b - reserve : free_b / 2;
        fit = usable / 10 * 9 / 16;
    }
    fit = env_u64_host("KT_CTR_MAX_SLOTS", fit);  // (tests: force the out-of-core passes)
    if (fit < 1024) fit = 1024;
    passes_ = (uint32_t)((want + fit - 1) / fit);
    if (passes_ < 1) passes_ = 1;
    // a partition's share of the keys varies a little: 1 / passes + 5 sigma of room.  The library rounds the request up
    // (to m * 2^j, at most 1.25x), so a table sized to the edge of the free HBM may not fit after all: more passes then.
    uint64_t cap = 0;
    for (;; passes_++) {
        cap = passes_ == 1 ? want : want / passes_ + want / passes_ / 16 + 4096;
        const int rc = kt_ctr_create(dev_.ctx, ksize_, cap, &ctr_);
        if (rc == KT_OK) break;
        if (rc != KT_ERR_NOMEM || passes_ >= 4096) return kt_last_error();
    }
    if (getenv("KT_CLI_TIMING")) {
        uint64_t slots = cap;
        (void)kt_ctr_capacity(ctr_, &slots);
        fprintf(stderr, "[timing] ctr setup: sizing (pre-pass only for compressed input) %.3f s, device init %.3f s, table of %llu slots, %u pass(es) %.3f s\n",
                t_stats, t_dev, (unsigned long long)slots, passes_, setup());
    }
    FILE *out = nullptr;
    const std::string path = out_dir_ + "/kmers.counts";
    PhaseTimer pt("ctr count (after the seq_stats pre-pass)");
    const bool timing = getenv("KT_CLI_TIMING") != nullptr;
    // One reader, one batch and one table writer serve every pass: a pass rewinds the reader (its threads, cuts and
    // buffers stay) and reuses the writer's slab, so after the first pass the loop allocates nothing - the host's memory
    // is the same whether the table takes 2 passes or 200 (the reference's ceiling: counter/src/lib.rs:114-118, 220-230).
    SeqReader reader;
    if (!reader.open(in_path_, false)) return reader.error();
    Batch b;
    TableWriter writer(acgt_, ksize_, threads_, memory_ceil_gb_);
    uint64_t heap_after_first = 0, rss_after_first = 0;
    for (uint32_t pass = 0; pass < passes_; pass++) {
        if (pass && !reader.rewind()) return reader.error();
        Lap lap;
        for (;;) {
            const bool more = reader.next_batch(b, cli_batch_bases(batch_bases), cli_batch_reads(1ull << 22));
            pt.t[0] += lap();
            if (b.n_reads() && !b.bases.empty()) {
                if (kt_ctr_add_reads_part(ctr_, b.bases.data(), b.offsets.data(), b.n_reads(), KT_MEM_HOST, passes_, pass) != KT_OK)
                    return kt_last_error();
            }
            pt.t[1] += lap();
            if (!more) break;
        }
        if (reader.failed()) return reader.error();
        if (passes_ == 1) break;  // the table stays resident: merge() (and cov) read it
        // out of core: this partition is complete - whoever needs to look k-mers up does it now (cov) -
        if (pass_hook_)
            if (std::string e = pass_hook_(pass, passes_, ctr_); !e.empty()) return e;
        // - its lines go to kmers.counts, and the table is reused
        if (!out) out = fopen(path.c_str(), "wb");
        if (!out) return "Unable to write to file: " + path;
        uint64_t n_pass = 0;
        if (std::string e = writer.write(out, ctr_, &n_pass); !e.empty()) return e;
        pt.t[3] += lap();
        if (timing) {
            // the ceiling's books, per pass: the program's own buffers by their capacities, the allocator's total, and
            // the kernel's view of the process (which includes the HIP runtime's ~0.6 GB)
            const MemReport r = mem_report();
            const uint64_t own = reader.buffer_bytes() + b.bases.capacity() + b.offsets.capacity() * 8 + writer.buffer_bytes();
            if (pass == 0) heap_after_first = r.heap_kb, rss_after_first = r.rss_kb;
            fprintf(stderr, "[timing] pass %u written: entries %llu, own buffers %llu kB, malloc in use %llu kB (peak at a slab boundary %llu kB), "
                            "malloc free %llu kB, VmRSS %llu kB (anon %llu, file %llu, shmem %llu), VmHWM %llu kB\n",
                    pass, (unsigned long long)n_pass, (unsigned long long)(own >> 10), (unsigned long long)r.heap_kb,
                    (unsigned long long)writer.heap_peak_kb(), (unsigned long long)r.heap_free_kb, (unsigned long long)r.rss_kb,
                    (unsigned long long)r.anon_kb, (unsigned long long)r.file_kb, (unsigned long long)r.shmem_kb,
                    (unsigned long long)r.hwm_kb);
        }
        if (kt_ctr_clear(ctr_) != KT_OK) return kt_last_error();
    }
    if (out) fclose(out);
    if (timing && passes_ > 1) {
        const MemReport r = mem_report();
        fprintf(stderr, "[timing] ctr out of core: %u passes, slabs of %llu entries, malloc peak at a slab boundary %llu kB, growth since the end "
                        "of pass 0: malloc in use %lld kB, VmRSS %lld kB, peak host memory (VmHWM) %llu kB\n",
                passes_, (unsigned long long)writer.slab(), (unsigned long long)writer.heap_peak_kb(),
                (long long)r.heap_kb - (long long)heap_after_first, (long long)r.rss_kb - (long long)rss_after_first,
                (unsigned long long)r.hwm_kb);
    }
    return "";
}

// ---- several GPUs: the table sharded by hash prefix, one worker thread (= one rank) per device -----------------
namespace {
// host all-to-all between the rank threads of one process (KT_CLI_SHARE_GPU=1: every rank on one GPU, where RCCL
// refuses to run - tests): everybody publishes its buffers, waits, copies its blocks, waits again
struct LocalFabric {
    int n;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    std::vector<const char *> send;
    explicit LocalFabric(int n_) : n(n_), send(n_) {}
    void barrier() {
        std::unique_lock<std::mutex> lk(m);
        const uint64_t g = gen;
        if (++arrived == n) {
            arrived = 0;
            gen++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g; });
        }
    }
};
struct FabricEnd { LocalFabric *f; int rank; };
int local_alltoall(void *user, const void *send, void *recv, uint64_t bytes) {
    FabricEnd *e = (FabricEnd *)user;
    e->f->send[e->rank] = (const char *)send;
    e->f->barrier();
    for (int p = 0; p < e->f->n; p++) memcpy((char *)recv + (uint64_t)p * bytes, e->f->send[p] + (uint64_t)e->rank * bytes, bytes);
    e->f->barrier();
    return 0;
}
}  // namespace

std::string CountComputer::count_sharded(uint64_t max_distinct) {
    const int N = n_devices_;
    const bool share = getenv("KT_CLI_SHARE_GPU") != nullptr;
    const uint64_t batch_bases = 256ull << 20, max_batch = batch_bases * 2;  // (a batch ends with a whole record)
    uint64_t per_rank = max_distinct / N + max_distinct / N / 16 + 4096;
    uint64_t cap = per_rank + per_rank / 10 * 9;
    uint8_t id[128] = {};
    if (!share) {
        // a rank thread whose device does not exist would leave the others waiting in ncclCommInitRank
        int n_dev = 0;
        if (kt_device_count(&n_dev) != KT_OK) return kt_last_error();
        if (dev_.index + N > n_dev)
            return "--devices " + std::to_string(N) + " from device " + std::to_string(dev_.index) + ": this node has " +
                   std::to_string(n_dev) + " GPU(s)";
        if (kt_rccl_unique_id(id) != KT_OK) return kt_last_error();
    }
    LocalFabric fabric(N);
    std::atomic<int> created_ok{0};
    const char *inject = getenv("KT_CLI_FAIL_RANK");  // tests: this rank's bring-up fails
    std::vector<FabricEnd> ends(N);
    release_shards();
    shards_.assign(N, nullptr);
    shard_ctx_.assign(N, nullptr);
    // rounds: the reader fills one batch per rank, the ranks add them collectively (a rank without a batch adds 0 reads)
    std::vector<Batch> batches[2];
    batches[0].resize(N);
    batches[1].resize(N);
    std::mutex m;
    std::condition_variable cv;
    int ready_round = -1, done_count[2] = {0, 0};
    bool last_round[2] = {false, false};
    std::vector<std::string> errs(N);
    auto worker = [&](int rank) {
        kt_ctx *ctx = nullptr;
        kt_sharded *sh = nullptr;
        auto fail = [&](const std::string &e) { errs[rank] = e.empty() ? "error" : e; };
        // bring-up in two steps with an agreement in between: everything a rank can fail at on its own (its device, its
        // allocations) happens first; only when EVERY rank got through does anyone enter a collective - the communicator's
        // creation, then the counting.  Otherwise all ranks stop here with the error (no rank is left waiting for one
        // that never comes).
        int rc = kt_ctx_create(share ? dev_.index : dev_.index + rank, nullptr, 1, &ctx);
        if (rc == KT_OK) rc = kt_sharded_create_local(ctx, ksize_, cap, max_batch, N, rank, &sh);
        const bool injected = rc == KT_OK && inject && atoi(inject) == rank;
        if (rc == KT_OK && !injected) created_ok.fetch_add(1);
        else fail(injected ? "injected bring-up failure (KT_CLI_FAIL_RANK)" : kt_last_error());
        fabric.barrier();
        const bool all_up = created_ok.load() == N;
        if (all_up) {
            ends[rank] = FabricEnd{&fabric, rank};
            rc = share ? kt_sharded_connect_host(sh, local_alltoall, &ends[rank]) : kt_sharded_connect_rccl(sh, id);
            if (rc != KT_OK) fail(kt_last_error());  // (a communicator that fails on one rank fails on its peers too)
        } else {
            if (errs[rank].empty()) fail("another device's bring-up failed");
            if (sh) kt_sharded_destroy(sh);
            sh = nullptr;
        }
        for (int round = 0;; round++) {
            bool last;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return ready_round >= round; });
                last = last_round[round & 1];
            }
            Batch &b = batches[round & 1][rank];
            // every rank takes part in every collective call, failed or not (the others would wait for ever)
            if (sh) {
                const uint64_t n = b.bases.empty() ? 0 : b.n_reads();
                if (kt_sharded_add_reads(sh, b.bases.data(), b.offsets.data(), n, KT_MEM_HOST) != KT_OK && errs[rank].empty())
                    fail(kt_last_error());
            }
            {
                std::lock_guard<std::mutex> lk(m);
                done_count[round & 1]++;
            }
            cv.notify_all();
            if (last) break;
        }
        if (sh) {
            if (kt_sharded_finalize(sh) != KT_OK && errs[rank].empty()) fail(kt_last_error());
            kt_ctr *t = nullptr;
            uint64_t n = 0;
            if (errs[rank].empty() && (kt_sharded_table(sh, &t) != KT_OK || kt_ctr_size(t, &n) != KT_OK)) fail(kt_last_error());
        }
        // the shard stays where it is: merge() writes it out in slabs (the reference streams its partitions into the file
        // the same way, counter/src/lib.rs:220-230) - no rank's export is ever held in host memory as a whole
        shards_[rank] = sh;
        shard_ctx_[rank] = ctx;
    };
    std::vector<std::thread> threads;
    for (int r = 0; r < N; r++) threads.emplace_back(worker, r);
    SeqReader reader;
    std::string err;
    if (!reader.open(in_path_, false)) err = reader.error();
    bool more = err.empty();
    for (int round = 0;; round++) {
        {   // the buffers of this parity are free once the round that used them last has been processed
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return round < 2 || done_count[round & 1] == N; });
            done_count[round & 1] = 0;
        }
        for (int r = 0; r < N; r++) {
            Batch &b = batches[round & 1][r];
            b.clear();
            if (more) more = reader.next_batch(b, cli_batch_bases(batch_bases), cli_batch_reads(1ull << 22));
        }
        if (reader.failed() && err.empty()) err = reader.error();
        {
            std::lock_guard<std::mutex> lk(m);
            last_round[round & 1] = !more;
            ready_round = round;
        }
        cv.notify_all();
        if (!more) break;
    }
    for (auto &t : threads) t.join();
    for (const auto &e : errs)
        if (!e.empty() && err.empty()) err = e;
    sharded_done_ = err.empty();
    return err;
}

std::string CountComputer::merge(bool /*del: no temp files exist to delete*/) {
    if (passes_ > 1) return "";  // out of core: count() wrote every partition's lines as it completed
    const std::string path = out_dir_ + "/kmers.counts";
    if (sharded_done_) {  // the shards, one after the other, each in slabs (the reference's line order is unspecified)
        FILE *out = fopen(path.c_str(), "wb");
        if (!out) return "Unable to write to file: " + path;
        std::string err;
        TableWriter writer(acgt_, ksize_, threads_, memory_ceil_gb_);
        for (size_t r = 0; r < shards_.size() && err.empty(); r++) {
            kt_ctr *t = nullptr;
            if (kt_sharded_table(shards_[r], &t) != KT_OK) err = kt_last_error();
            else err = writer.write(out, t, nullptr);
        }
        fclose(out);
        if (getenv("KT_CLI_TIMING"))
            fprintf(stderr, "[timing] ctr --devices %d written: VmHWM %llu kB\n", n_devices_, (unsigned long long)peak_rss_kb());
        if (!keep_shards_) release_shards();
        return err;
    }
    if (!ctr_) return "count() has not run";
    PhaseTimer pt("ctr merge");
    Lap lap;
    FILE *out = fopen(path.c_str(), "wb");
    if (!out) return "Unable to write to file: " + path;
    TableWriter writer(acgt_, ksize_, threads_, memory_ceil_gb_);
    const std::string e = writer.write(out, ctr_, nullptr);
    pt.t[2] += lap();
    fclose(out);
    pt.t[3] += lap();
    if (getenv("KT_CLI_TIMING")) fprintf(stderr, "[timing] ctr merge: peak host memory (VmHWM) %llu kB\n", (unsigned long long)peak_rss_kb());
    return e;
}

// ---------------------------------------------------------------------------------------------
CovComputer::CovComputer(std::string in_path, std::string out_dir, int ksize, uint64_t bin_size, uint64_t bin_count)
    : in_path_(in_path), in_path_kmer_(std::move(in_path)), out_dir_(std::move(out_dir)), ksize_(ksize),
      bin_size_(bin_size), bin_count_(bin_count) {}

CovComputer::~CovComputer() { delete ctr_; }

// one pass of an out-of-core table is complete: the raw bin counts of every read's k-mers of this hash partition
std::string CovComputer::cov_pass(uint32_t pass, uint32_t passes, kt_ctr *table) {
    SeqReader reader;
    if (!reader.open(in_path_, false)) return reader.error();
    Batch b;
    uint64_t at = 0;
    for (;;) {
        const bool more = reader.next_batch(b, cli_batch_bases(256ull << 20), cli_batch_reads(bin_count_ >= 2048 ? 8192 : 1ull << 19));
        const uint64_t n = b.n_reads();
        if (n) {
            if (pass == 0) acc_rows_.resize((at + n) * bin_count_, 0u);
            else if ((at + n) * bin_count_ > acc_rows_.size()) return "cov: the input changed between the passes";
            if (kt_cov_batch_part(table, bases_ptr(b), b.offsets.data(), n, bin_size_, bin_count_, acc_rows_.data() + at * bin_count_,
                                  KT_MEM_HOST, passes, pass) != KT_OK)
                return kt_last_error();
            at += n;
        }
        if (!more) break;
    }
    if (reader.failed()) return reader.error();
    acc_reads_ = at;
    return "";
}

std::string CovComputer::build_table() {
    delete ctr_;
    acc_rows_.clear();
    acc_reads_ = 0;
    ctr_ = new CountComputer(in_path_kmer_, out_dir_, ksize_);
    ctr_->set_threads(threads_);
    ctr_->set_max_memory(memory_ceil_gb_);
    ctr_->set_device(device_);
    ctr_->set_devices(n_devices_);
    ctr_->set_keep_shards(true);
    ctr_->set_pass_hook([this](uint32_t pass, uint32_t passes, kt_ctr *t) { return cov_pass(pass, passes, t); });
    std::string e = ctr_->count();
    if (e.empty()) e = ctr_->merge(true);  // the reference leaves kmers.counts behind as well
    return e;
}

std::string CovComputer::compute_coverages() {
    // the reference parses kmers.counts back into a HashMap (:82-92); the table is still in HBM here
    if (ctr_ && ctr_->passes() > 1) {
        // ... unless it took several passes: the rows were summed pass by pass (cov_pass); normalise, format, write
        const std::string path = out_dir_ + "/kmers.vectors";
        FILE *out = fopen(path.c_str(), "wb");
        if (!out) return "Unable to write to file: " + path;
        PhaseTimer pt("cov (rows summed over the passes)");
        AsyncWriter writer(out, pt);
        std::vector<std::string> pieces;
        const uint64_t bins = bin_count_, slab = bins >= 2048 ? 8192 : 1ull << 19;
        Work w;
        for (uint64_t r0 = 0; r0 < acc_reads_; r0 += slab) {
            const uint64_t n = acc_reads_ - r0 < slab ? acc_reads_ - r0 : slab;
            w.b.clear();
            w.b.offsets.assign(n + 1, 0);  // (emit_matrix only needs the number of reads)
            w.resize_rows(n * bins);
            for (uint64_t r = 0; r < n; r++) {
                const uint32_t *row = acc_rows_.data() + (r0 + r) * bins;
                uint64_t total = 0;
                for (uint64_t i = 0; i < bins; i++) total += row[i];
                const double d = total > 1 ? (double)total : 1.0;   // coverage/src/lib.rs:180-182
                for (uint64_t i = 0; i < bins; i++) w.rows[r * bins + i] = norm_ ? (double)row[i] / d : (double)row[i];
            }
            emit_matrix(writer, w, bins, norm_, delim_, threads_, pieces, pt);
        }
        writer.finish();
        fclose(out);
        return "";
    }
    if (ctr_ && ctr_->n_shards() > 0) {
        // --devices N: the table is N shards on N GPUs, each holding the k-mers whose minimiser it owns.  Every
        // batch of reads goes past every shard - one thread per shard, each adding raw bin counts to rows of its own
        // (kt_cov_batch_part: a k-mer a shard does not hold may be on another) -, the shards' rows are summed, normalised with one
        // division per cell and written: every k-mer of every read has been binned exactly once (the reference looks the
        // k-mers up in one map, coverage/src/lib.rs:165-184; its CountComputer built that map out of partitions the same way)
        const size_t N = ctr_->n_shards();
        SeqReader reader;
        if (!reader.open(in_path_, false)) return reader.error();
        const std::string path = out_dir_ + "/kmers.vectors";
        FILE *out = fopen(path.c_str(), "wb");
        if (!out) return "Unable to write to file: " + path;
        PhaseTimer pt("cov (rows summed over the shards)");
        AsyncWriter writer(out, pt);
        std::vector<std::string> pieces;
        const uint64_t bins = bin_count_;
        std::vector<std::vector<uint32_t>> part(N);
        std::vector<std::string> errs(N);
        Work w;
        std::string err;
        for (;;) {
            // (a batch holds N shards' u32 rows, the library's temporary of one shard's rows and the f64 rows: its reads are
            // bounded so that all of it stays under 1 GiB whatever the number of shards and bins - ADVICE r5)
            const uint64_t by_memory = std::max<uint64_t>(1024, (1ull << 30) / (bins * (4 * (uint64_t)N + 12)));
            const bool more = reader.next_batch(w.b, cli_batch_bases(256ull << 20),
                                                std::min<uint64_t>(cli_batch_reads(bins >= 2048 ? 8192 : 1ull << 19), by_memory));
            const uint64_t n = w.b.n_reads();
            if (n) {
                std::vector<std::thread> th;
                for (size_t r = 0; r < N; r++)
                    th.emplace_back([&, r] {
                        part[r].assign((size_t)(n * bins), 0u);
                        kt_ctr *t = ctr_->shard_table(r);
                        if (!t || kt_cov_batch_part(t, bases_ptr(w.b), w.b.offsets.data(), n, bin_size_, bin_count_, part[r].data(),
                                                    KT_MEM_HOST, 1, 0) != KT_OK)
                            errs[r] = t ? kt_last_error() : "cov: a shard's table is gone";
                    });
                for (auto &t : th) t.join();
                for (const auto &e : errs)
                    if (!e.empty() && err.empty()) err = e;
                if (!err.empty()) break;
                w.resize_rows(n * bins);
                for (uint64_t r = 0; r < n; r++) {
                    uint64_t total = 0;
                    for (uint64_t i = 0; i < bins; i++) {
                        uint32_t c = 0;  // (modulo 2^32: a shard's own cells mean nothing before they are summed - kt_cov_batch_part)
                        for (size_t sh = 0; sh < N; sh++) c += part[sh][r * bins + i];
                        w.rows[r * bins + i] = (double)c;
                        total += c;
                    }
                    if (norm_) {
                        const double d = total > 1 ? (double)total : 1.0;  // coverage/src/lib.rs:180-182
                        for (uint64_t i = 0; i < bins; i++) w.rows[r * bins + i] /= d;
                    }
                }
                emit_matrix(writer, w, bins, norm_, delim_, threads_, pieces, pt);
            }
            if (!more) break;
        }
        writer.finish();
        fclose(out);
        if (err.empty() && reader.failed()) err = reader.error();
        return err;
    }
    if (!ctr_ || !ctr_->table()) return "build_table() has not run";
    SeqReader reader;
    if (!reader.open(in_path_, false)) return reader.error();
    const std::string path = out_dir_ + "/kmers.vectors";
    FILE *out = fopen(path.c_str(), "wb");
    if (!out) return "Unable to write to file: " + path;
    std::vector<std::string> pieces;
    const uint64_t bins = bin_count_;
    PhaseTimer pt("cov");
    AsyncWriter writer(out, pt);
    const std::string err = run_pipeline(
        reader, batch_bases(256ull << 20), bins >= 2048 ? 8192 : 1ull << 19, pt,
        [&](Work &w) -> std::string {
            const uint64_t n = w.b.n_reads();
            w.resize_rows(n * bins);
            w.pin_rows(ctr_->context());
            if (kt_cov_batch(ctr_->table(), bases_ptr(w.b), w.b.offsets.data(), n, bin_size_, bin_count_, norm_, KT_F64,
                             w.rows.data(), KT_MEM_HOST) != KT_OK)
                return kt_last_error();
            return "";
        },
        [&](Work &w) { emit_matrix(writer, w, bins, norm_, delim_, threads_, pieces, pt); });  // :112-124
    writer.finish();
    fclose(out);
    return err;
}

// ---------------------------------------------------------------------------------------------
// minimisers: one C-ABI call per batch; the capacity is a guess that is corrected on the first miss
static std::string minimiser_batch(kt_ctx *ctx, Work &w, uint64_t wsize, int msize) {
    const uint64_t n = w.b.n_reads();
    w.u64[0].resize(n + 1);
    uint64_t cap = w.u64[1].size();
    if (cap < n + 16) cap = wsize ? w.b.bases.size() / 4 + n + 16 : n + 16;
    for (int attempt = 0; attempt < 2; attempt++) {
        for (int i = 1; i < 4; i++) w.u64[i].resize(cap);
        const int rc = kt_minimisers(ctx, bases_ptr(w.b), w.b.offsets.data(), n, wsize, msize, w.u64[0].data(),
                                     w.u64[1].data(), w.u64[2].data(), w.u64[3].data(), cap, &w.n_events, KT_MEM_HOST);
        if (rc == KT_OK) return "";
        if (w.n_events <= cap) return kt_last_error();  // a real error, not a capacity miss
        cap = w.n_events;
    }
    return kt_last_error();
}

static void append_mmer(std::string &s, uint64_t kmer, int msize) {
    char buf[40];
    kt_numeric_to_kmer(kmer, msize, buf);  // u64::MAX (no full window) prints as all T, like the reference
    s += buf;
}

static void append_u64(std::string &s, uint64_t v) {
    char buf[24];
    const auto r = std::to_chars(buf, buf + sizeof buf, v);
    s.append(buf, (size_t)(r.ptr - buf));
}

static std::string check_min_args(uint64_t wsize, int msize, const std::string &in_path) {
    if (msize < 1 || msize > 31) return "minimiser size must be in 1..31";
    if (wsize != 0 && wsize < (uint64_t)msize) return "window size must not be shorter than the minimiser";
    if (format_from_path(in_path) == SeqFormat::Unknown && in_path != "-")
        return "unsupported input extension (expected .fa/.fasta/.fna/.fq/.fastq[.gz]): " + in_path;  // SeqFormat::get().unwrap()
    return "";
}

std::string seq_to_min(uint64_t wsize, int msize, const std::string &in_path, const std::string &out_path, int threads,
                       int device) {
    if (std::string e = check_min_args(wsize, msize, in_path); !e.empty()) return e;
    SeqReader reader;
    if (!reader.open(in_path, false)) return reader.error();
    FILE *out = fopen(out_path.c_str(), "wb");
    if (!out) return "Unable to write to file: " + out_path;
    Device dev;
    dev.index = device;
    if (std::string e = dev.ensure(); !e.empty()) {
        fclose(out);
        return e;
    }
    std::vector<std::string> pieces;
    PhaseTimer pt("min s2m");
    AsyncWriter writer(out, pt);
    const std::string err = run_pipeline(
        reader, 64ull << 20, 1ull << 19, pt, [&](Work &w) { return minimiser_batch(dev.ctx, w, wsize, msize); },
        [&](Work &w) {
            Lap lap;
            const uint64_t *evo = w.u64[0].data(), *km = w.u64[1].data(), *st = w.u64[2].data(), *en = w.u64[3].data();
            const uint64_t n = w.b.n_reads();
            const size_t per_row = n ? (size_t)(w.n_events / n + 1) * (size_t)(msize + 12) + 24 : 1;
            format_rows(n, threads, per_row, pieces, [&](uint64_t r, std::string &s) {
                s += w.b.ids[r];  // mins.join("\t") over [id, "MMER:s-e"..., "\n"]  (:131-141)
                for (uint64_t j = evo[r]; j < evo[r + 1]; j++) {
                    s += '\t';
                    append_mmer(s, km[j], msize);
                    s += ':';
                    append_u64(s, st[j]);
                    s += '-';
                    append_u64(s, en[j]);
                }
                s += "\t\n";
            });
            pt.t[2] += lap();
            writer.write(pieces);
        },
        true);
    writer.finish();
    fclose(out);
    return err;
}

// Rust's {:?} for a String: quotes, with \" \\ \t \n \r and other control characters escaped
static void append_debug_str(std::string &s, const std::string &id) {
    s += '"';
    for (unsigned char c : id) {
        switch (c) {
            case '"': s += "\\\""; break;
            case '\\': s += "\\\\"; break;
            case '\t': s += "\\t"; break;
            case '\n': s += "\\n"; break;
            case '\r': s += "\\r"; break;
            case '\'': s += "'"; break;
            default:
                if (c < 0x20 || c == 0x7f) {
                    char buf[16];
                    snprintf(buf, sizeof buf, "\\u{%x}", c);
                    s += buf;
                } else {
                    s += (char)c;
                }
        }
    }
    s += '"';
}

std::string bin_sequences(uint64_t wsize, int msize, const std::string &in_path, const std::string &out_path, int threads,
                          int device) {
    if (std::string e = check_min_args(wsize, msize, in_path); !e.empty()) return e;
    SeqReader reader;
    if (!reader.open(in_path, false)) return reader.error();
    FILE *out = fopen(out_path.c_str(), "wb");
    if (!out) return "Unable to write to file: " + out_path;
    Device dev;
    dev.index = device;
    if (std::string e = dev.ensure(); !e.empty()) {
        fclose(out);
        return e;
    }
    // the whole file's triples are grouped by minimiser at the end, like the reference's in-memory map
    struct Hit {
        uint64_t kmer;
        uint32_t id, start, end;  // id = index into `ids`
    };
    std::vector<Hit> hits;
    std::vector<std::string> ids;
    PhaseTimer pt("min m2s");
    const std::string err = run_pipeline(
        reader, 64ull << 20, 1ull << 19, pt, [&](Work &w) { return minimiser_batch(dev.ctx, w, wsize, msize); },
        [&](Work &w) {
            Lap lap;
            const uint64_t *evo = w.u64[0].data();
            const uint32_t id0 = (uint32_t)ids.size();
            ids.insert(ids.end(), w.b.ids.begin(), w.b.ids.end());
            for (uint64_t r = 0; r < w.b.n_reads(); r++)
                for (uint64_t j = evo[r]; j < evo[r + 1]; j++)
                    hits.push_back(Hit{w.u64[1][j], id0 + (uint32_t)r, (uint32_t)w.u64[2][j], (uint32_t)w.u64[3][j]});
            pt.t[2] += lap();
        },
        true);
    if (!err.empty()) {
        fclose(out);
        return err;
    }
    Lap lap;
    // ascending minimiser text == ascending value of its low 2m bits (A < C < G < T is the 2-bit order)
    const uint64_t mask = msize >= 32 ? ~0ull : ((1ull << (2 * msize)) - 1ull);
    std::stable_sort(hits.begin(), hits.end(), [&](const Hit &x, const Hit &y) { return (x.kmer & mask) < (y.kmer & mask); });
    std::vector<uint64_t> group_start;
    for (uint64_t i = 0; i < hits.size(); i++)
        if (i == 0 || (hits[i].kmer & mask) != (hits[i - 1].kmer & mask)) group_start.push_back(i);
    group_start.push_back(hits.size());
    std::vector<std::string> pieces;
    const uint64_t n_groups = group_start.size() - 1;
    format_rows(n_groups, threads, (size_t)msize + 48, pieces, [&](uint64_t g, std::string &s) {
        append_mmer(s, hits[group_start[g]].kmer, msize);  // "{k}\t{v:?}\n"  (:81-84)
        s += "\t[";
        for (uint64_t i = group_start[g]; i < group_start[g + 1]; i++) {
            if (i != group_start[g]) s += ", ";
            s += '(';
            append_debug_str(s, ids[hits[i].id]);
            s += ", ";
            append_u64(s, hits[i].start);
            s += ", ";
            append_u64(s, hits[i].end);
            s += ')';
        }
        s += "]\n";
    });
    pt.t[2] += lap();
    for (const auto &p : pieces) fwrite(p.data(), 1, p.size(), out);
    fclose(out);
    pt.t[3] += lap();
    return "";
}

}  // namespace kthost
