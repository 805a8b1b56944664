#include "seqio.hpp"

#include <ctype.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace kthost {

static bool ends_with(const std::string &s, const char *suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

SeqFormat format_from_path(const std::string &path_in) {
    std::string path = path_in;
    // the reference strips every trailing ".gz" (trim_end_matches), then tests the suffix
    if (ends_with(path, ".gz"))
        while (ends_with(path, ".gz")) path.resize(path.size() - 3);
    if (ends_with(path, ".fq") || ends_with(path, ".fastq")) return SeqFormat::Fastq;
    if (ends_with(path, ".fasta") || ends_with(path, ".fa") || ends_with(path, ".fna")) return SeqFormat::Fasta;
    return SeqFormat::Unknown;
}

// ---- parallel parse of a mapped plain file -----------------------------------------------------------------------
void SeqReader::open_range(const unsigned char *p, size_t n, SeqFormat fmt) {
    err_.clear();
    base_ = p;
    mem_ = true;
    pos_ = 0;
    end_ = n;
    eof_ = false;
    fmt_ = fmt;
    have_pending_ = false;
    n_records_ = 0;
}

struct SeqReader::Parallel {
    int fd = -1;       // pieces are pread() into per-worker buffers (mapping the file made the parsers' page faults
    size_t size = 0;   // and the HIP runtime's pinning of pageable copies fight over the address space lock)
    SeqFormat fmt = SeqFormat::Fasta;
    std::vector<size_t> cuts;  // piece i = [cuts[i], cuts[i + 1])
    uint64_t max_bases = 0, max_reads = 0;
    bool keep_ids = false, started = false;
    // pieces are parsed by `workers` threads in order of their index; a piece's batches wait in `done` until the
    // consumer has taken every earlier piece (at most `window` pieces are parsed ahead of the consumer)
    struct Piece {
        std::vector<Batch> batches;
        std::string err;
        bool ready = false;
    };
    std::vector<Piece> pieces;
    std::mutex m;
    std::condition_variable cv;
    size_t next_piece = 0, consume_piece = 0, consume_batch = 0, window = 0;
    int busy = 0;                  // workers parsing a piece right now
    std::vector<size_t> buf_bytes = std::vector<size_t>(8, 0);  // (statistics: the workers' piece buffers)
    bool stop = false;
    std::vector<std::thread> workers;
    std::vector<Batch> spare;  // buffers of batches the consumer has taken, for the workers to fill again
    uint64_t records = 0;

    ~Parallel() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : workers) t.join();
        if (fd >= 0) ::close(fd);
    }
    bool read_at(size_t at, size_t n, std::vector<unsigned char> &buf) const {
        buf.resize(n);
        size_t got = 0;
        while (got < n) {
            const ssize_t r = pread(fd, buf.data() + got, n - got, (off_t)(at + got));
            if (r <= 0) return false;
            got += (size_t)r;
        }
        return true;
    }

    // first record start at or after `from` (FASTA: a line that begins with '>'; FASTQ: a line that begins with '@'
    // whose third line begins with '+' and whose fourth is as long as its second - two such records in a row).
    // Looks at a window of the file; a record start is always found within it for records below ~8 MB.
    size_t boundary(size_t from) const {
        if (from == 0) return 0;
        // (a window that grows: records are a few hundred bytes as a rule, and a reader is opened once per pass of an
        // out-of-core count - sixteen megabytes per cut were most of what such a pass read beside the pieces)
        std::vector<unsigned char> win;
        for (size_t wmax = 64u << 10;; wmax *= 16) {
            const size_t w0 = from - 1, wn = size - w0 < wmax ? size - w0 : wmax;
            if (!read_at(w0, wn, win)) return size;
            const bool to_eof = w0 + wn == size;
            const size_t c = boundary_in(win.data(), wn, w0, to_eof);
            if (c < size || to_eof || wmax >= (16u << 20)) return c;
        }
    }
    size_t boundary_in(const unsigned char *map, size_t wsize, size_t w0, bool to_eof) const {
        const unsigned char *nl = (const unsigned char *)memchr(map, '\n', wsize);
        if (!nl) return size;
        size_t p = (size_t)(nl - map) + 1;
        auto line_end = [&](size_t q) {
            const unsigned char *e = q < wsize ? (const unsigned char *)memchr(map + q, '\n', wsize - q) : nullptr;
            return e ? (size_t)(e - map) : wsize;
        };
        while (p < wsize) {
            if (fmt == SeqFormat::Fasta) {
                if (map[p] == '>') return w0 + p;
            } else if (map[p] == '@') {
                size_t q = p;
                bool ok = true;
                for (int rec = 0; rec < 2 && ok && q < wsize; rec++) {
                    if (map[q] != '@') { ok = false; break; }
                    const size_t e0 = line_end(q), s1 = e0 + 1, e1 = line_end(s1), s2 = e1 + 1, e2 = line_end(s2),
                                 s3 = e2 + 1, e3 = line_end(s3);
                    if (e3 >= wsize && !to_eof) { ok = false; break; }  // the window ended inside the record
                    if (s2 >= wsize || map[s2] != '+' || s3 > wsize) { ok = false; break; }
                    size_t l1 = e1 - s1, l3 = e3 > s3 ? e3 - s3 : 0;
                    if (l1 && map[s1 + l1 - 1] == '\r') l1--;
                    if (l3 && map[s3 + l3 - 1] == '\r') l3--;
                    if (l1 != l3) { ok = false; break; }
                    q = e3 + 1;
                }
                if (ok) return w0 + p;
            }
            p = line_end(p) + 1;
        }
        return size;
    }

    void work(int self) {
        std::vector<unsigned char> buf;  // this worker's piece of the file, reused
        for (;;) {
            size_t i;
            {
                std::unique_lock<std::mutex> lk(m);
                // (a worker that finds no piece left waits: rewind() hands the pieces out again)
                cv.wait(lk, [&] { return stop || (next_piece < pieces.size() && next_piece < consume_piece + window); });
                if (stop) return;
                i = next_piece++;
                busy++;
            }
            SeqReader r;
            std::vector<Batch> out;
            std::string err;
            if (!read_at(cuts[i], cuts[i + 1] - cuts[i], buf)) err = "read error";
            r.open_range(buf.data(), err.empty() ? buf.size() : 0, fmt);
            for (;;) {
                out.emplace_back();
                {
                    std::lock_guard<std::mutex> lk(m);
                    if (!spare.empty()) {
                        out.back().bases.swap(spare.back().bases);
                        out.back().offsets.swap(spare.back().offsets);
                        spare.pop_back();
                    }
                }
                const bool more = r.next_batch(out.back(), max_bases, max_reads, keep_ids);
                if (out.back().n_reads() == 0) out.pop_back();
                if (!more) break;
            }
            if (r.failed()) err = r.error();
            {
                std::lock_guard<std::mutex> lk(m);
                pieces[i].batches = std::move(out);
                pieces[i].err = std::move(err);
                pieces[i].ready = true;
                busy--;
                buf_bytes[(size_t)self % buf_bytes.size()] = buf.capacity();
            }
            cv.notify_all();
        }
    }
};

// The piecewise reader finds FASTQ record starts by the 4-line shape of a record (Parallel::boundary).  The serial
// parser also takes sequences and qualities wrapped over several lines; such a file must stay with it.  Looks at the
// first records of the file: '@' line, sequence, '+' line, quality of the sequence's length - up to 64 of them.
static bool fastq_is_four_line(int fd, size_t size) {
    std::vector<unsigned char> head(size < (1u << 20) ? size : (1u << 20));
    size_t got = 0;
    while (got < head.size()) {
        const ssize_t r = pread(fd, head.data() + got, head.size() - got, (off_t)got);
        if (r <= 0) return false;
        got += (size_t)r;
    }
    size_t p = 0;
    auto line = [&](size_t &s0, size_t &len) {  // next line [s0, s0 + len), CR stripped; false at the sample's end
        if (p >= got) return false;
        const unsigned char *e = (const unsigned char *)memchr(head.data() + p, '\n', got - p);
        if (!e) return false;
        s0 = p;
        len = (size_t)(e - head.data()) - p;
        if (len && head[s0 + len - 1] == '\r') len--;
        p = (size_t)(e - head.data()) + 1;
        return true;
    };
    for (int rec = 0; rec < 64; rec++) {
        size_t s0, l0, s1, l1, s2, l2, s3, l3;
        if (!line(s0, l0)) return rec > 0;          // (the sample ended between records: what was seen was regular)
        if (l0 == 0 || head[s0] != '@') return false;
        if (!line(s1, l1) || !line(s2, l2) || !line(s3, l3)) return rec > 0;
        if (l2 == 0 || head[s2] != '+' || l1 != l3) return false;
    }
    return true;
}

static int reader_threads() {
    if (const char *e = getenv("KT_READER_THREADS")) return atoi(e) > 0 ? atoi(e) : 1;
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // a container's CPU quota, not the host's thread count
        char q[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long c = (atol(q) + period - 1) / period;
            if (c > 0 && c < n) n = c;
        }
        fclose(f);
    }
    n = n / 4;        // a quarter of the CPUs: the formatters and the runtime's staging copies need the rest
    if (n > 4) n = 4;  // (four parsers deliver ~8 GB/s, more than any later stage takes)
    return n < 1 ? 1 : (int)n;
}

SeqReader::~SeqReader() {
    if (gz_) gzclose(gz_);
}

bool SeqReader::open(const std::string &path, bool sniff) {
    err_.clear();
    par_.reset();
    // a plain regular file of some size: map it and parse pieces of it concurrently
    if (path != "-" && reader_threads() > 1) {
        struct stat st;
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= (32 << 20)) {
            unsigned char magic[2] = {0, 0};
            const bool gz = pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
            if (!gz) {
                (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
                auto P = std::make_shared<Parallel>();
                P->fd = fd;
                P->size = (size_t)st.st_size;
                SeqFormat f = sniff ? SeqFormat::Unknown : format_from_path(path);
                if (f == SeqFormat::Unknown) f = magic[0] == '>' ? SeqFormat::Fasta : SeqFormat::Fastq;
                if (f != SeqFormat::Fastq || fastq_is_four_line(fd, (size_t)st.st_size)) {
                    P->fmt = f;
                    fmt_ = f;
                    par_ = P;
                    return true;
                }
                P->fd = -1;  // a wrapped FASTQ: the serial parser below (the descriptor is closed just after)
            }
        }
        if (fd >= 0) ::close(fd);
    }
    if (path == "-") {
        gz_ = gzdopen(0, "rb");
    } else {
        gz_ = gzopen(path.c_str(), "rb");  // transparent for non-gzip files
    }
    if (!gz_) {
        err_ = "Unable to open: " + path;
        return false;
    }
    gzbuffer(gz_, 1 << 20);
    buf_.resize(1 << 22);
    fmt_ = sniff ? SeqFormat::Unknown : format_from_path(path);
    if (fmt_ == SeqFormat::Unknown) {
        int c;
        if (!peek(c)) {
            if (!err_.empty()) return false;
            fmt_ = SeqFormat::Fasta;  // empty input: no records either way
        } else {
            fmt_ = (c == '>') ? SeqFormat::Fasta : SeqFormat::Fastq;
        }
    }
    return true;
}

// Back to the first record (the passes of an out-of-core count): the parallel reader keeps its cuts, its worker threads
// and every buffer it has - a second pass allocates nothing -, the stream reader rewinds its file.
bool SeqReader::rewind() {
    err_.clear();
    n_records_ = 0;
    if (par_) {
        Parallel &P = *par_;
        std::unique_lock<std::mutex> lk(P.m);
        if (!P.started) return true;
        const size_t n = P.pieces.size();
        P.next_piece = n;  // nobody takes a new piece while the ones in work are waited for
        P.cv.wait(lk, [&] { return P.busy == 0; });
        for (auto &pc : P.pieces) {
            for (auto &bt : pc.batches) {  // parsed ahead and never taken: the buffers go back to the pool
                P.spare.emplace_back();
                P.spare.back().bases.swap(bt.bases);
                P.spare.back().offsets.swap(bt.offsets);
            }
            pc.batches.clear();
            pc.err.clear();
            pc.ready = false;
        }
        P.next_piece = P.consume_piece = P.consume_batch = 0;
        P.records = 0;
        lk.unlock();
        P.cv.notify_all();
        return true;
    }
    if (!gz_ || gzrewind(gz_) != 0) {
        err_ = "Unable to rewind the input";
        return false;
    }
    pos_ = end_ = 0;
    eof_ = false;
    have_pending_ = false;
    return true;
}

// bytes held in the reader's own buffers right now (KT_CLI_TIMING: the memory ceiling's books)
size_t SeqReader::buffer_bytes() {
    size_t n = buf_.capacity() + line_.capacity() + pending_.capacity();
    if (par_) {
        Parallel &P = *par_;
        std::lock_guard<std::mutex> lk(P.m);
        for (const auto &pc : P.pieces)
            for (const auto &bt : pc.batches) n += bt.bases.capacity() + bt.offsets.capacity() * 8;
        for (const auto &bt : P.spare) n += bt.bases.capacity() + bt.offsets.capacity() * 8;
        for (const size_t b : P.buf_bytes) n += b;
    }
    return n;
}

bool SeqReader::fill() {
    if (eof_) return false;
    if (mem_) {  // a mapped range has been handed over whole
        eof_ = true;
        return false;
    }
    const int n = gzread(gz_, buf_.data(), (unsigned)buf_.size());
    if (n < 0) {
        int e = 0;
        err_ = std::string("read error: ") + gzerror(gz_, &e);
        eof_ = true;
        return false;
    }
    if (n == 0) {
        eof_ = true;
        return false;
    }
    base_ = buf_.data();
    pos_ = 0;
    end_ = (size_t)n;
    return true;
}

bool SeqReader::peek(int &c) {
    if (pos_ == end_ && !fill()) return false;
    c = base_[pos_];
    return true;
}

bool SeqReader::read_line(std::string &line) {
    line.clear();
    bool any = false;
    for (;;) {
        if (pos_ == end_ && !fill()) return any;
        any = true;
        const unsigned char *p = base_ + pos_;
        const unsigned char *nl = (const unsigned char *)memchr(p, '\n', end_ - pos_);
        if (nl) {
            line.append((const char *)p, (size_t)(nl - p));
            pos_ += (size_t)(nl - p) + 1;
            return true;
        }
        line.append((const char *)p, end_ - pos_);
        pos_ = end_;
    }
}

bool SeqReader::read_line_view(const char *&p, size_t &n) {
    if (pos_ == end_ && !fill()) return false;
    const unsigned char *b = base_ + pos_;
    const unsigned char *nl = (const unsigned char *)memchr(b, '\n', end_ - pos_);
    if (nl) {  // the common case: the line ends inside the buffer
        p = (const char *)b;
        n = (size_t)(nl - b);
        pos_ += n + 1;
        return true;
    }
    if (!read_line(line_)) return false;  // spans a refill (or is the unterminated last line): assemble a copy
    p = line_.data();
    n = line_.size();
    return true;
}

static void trim_end_view(const char *p, size_t &n) {
    while (n) {
        const unsigned char c = (unsigned char)p[n - 1];
        if (c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f') n--;
        else break;
    }
}

static void trim_end(std::string &s) {
    while (!s.empty()) {
        const unsigned char c = (unsigned char)s.back();
        if (c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f') s.pop_back();
        else break;
    }
}

static std::string first_token(const std::string &hdr) {
    size_t i = 1;  // skip '>' / '@'
    size_t j = i;
    while (j < hdr.size() && !isspace((unsigned char)hdr[j])) j++;
    return hdr.substr(i, j - i);
}

bool SeqReader::next_batch_parallel(Batch &b, bool keep_ids) {
    Parallel &P = *par_;
    if (!P.started) {
        P.started = true;
        P.keep_ids = keep_ids;
        // one piece should parse into one batch of the caller's size (small batches cost every later stage a call, a
        // thread start, a synchronisation): bytes per record and the share of sequence bytes from the head of the file
        {
            const size_t head = P.size < (4u << 20) ? P.size : (4u << 20);
            std::vector<unsigned char> hb;
            size_t lines = 0, recs = 0;
            if (P.read_at(0, head, hb)) {
                for (size_t i = 0; i < head; i++) {
                    if (hb[i] == '\n') lines++;
                    if (P.fmt == SeqFormat::Fasta && hb[i] == '>' && (i == 0 || hb[i - 1] == '\n')) recs++;
                }
            }
            if (P.fmt == SeqFormat::Fastq) recs = lines / 4;
            if (recs < 1) recs = 1;
            const double bytes_per_rec = (double)head / (double)recs;
            const double seq_share = P.fmt == SeqFormat::Fastq ? 0.48 : 0.9;
            double piece_d = 0.9 * (double)P.max_reads * bytes_per_rec;
            const double by_bases = 0.9 * (double)P.max_bases / seq_share;
            if (by_bases < piece_d) piece_d = by_bases;
            size_t piece = piece_d > 1e12 ? (size_t)1e12 : (size_t)piece_d;
            if (piece < (8u << 20)) piece = 8u << 20;
            P.cuts.push_back(0);
            for (size_t at = piece; at < P.size; at += piece) {
                const size_t c = P.boundary(at);
                if (c > P.cuts.back() && c < P.size) P.cuts.push_back(c);
            }
            P.cuts.push_back(P.size);
            P.pieces.resize(P.cuts.size() - 1);
        }
        const int T = reader_threads();
        P.window = (size_t)T + 2;
        for (int t = 0; t < T; t++) P.workers.emplace_back([&P, t] { P.work(t); });
    }
    b.clear();
    b.first_record = P.records;
    std::unique_lock<std::mutex> lk(P.m);
    for (;;) {
        if (P.consume_piece >= P.pieces.size()) return false;  // end of input
        Parallel::Piece &pc = P.pieces[P.consume_piece];
        P.cv.wait(lk, [&] { return pc.ready; });
        if (P.consume_batch < pc.batches.size()) {
            Batch &src = pc.batches[P.consume_batch++];
            // copied, not swapped: the caller's buffers keep their addresses from batch to batch (the HIP runtime
            // moves pageable memory it has seen before markedly faster, and page-locked callers stay page-locked);
            // the piece's buffers go back to the workers
            lk.unlock();
            b.bases.assign(src.bases.begin(), src.bases.end());
            b.offsets.assign(src.offsets.begin(), src.offsets.end());
            b.ids.swap(src.ids);
            lk.lock();
            P.spare.emplace_back();
            P.spare.back().bases.swap(src.bases);
            P.spare.back().offsets.swap(src.offsets);
            b.first_record = P.records;
            P.records += b.n_reads();
            n_records_ = P.records;
            // is anything left after this batch?  (next_batch returns false with the last records of the input)
            bool more = P.consume_batch < pc.batches.size() || !pc.err.empty();
            for (size_t j = P.consume_piece + 1; !more && j < P.pieces.size(); j++) more = true;
            if (more) return true;
            std::vector<Batch>().swap(pc.batches);
            P.consume_piece++;
            return false;
        }
        if (!pc.err.empty()) {
            err_ = pc.err;
            return false;
        }
        std::vector<Batch>().swap(pc.batches);  // piece exhausted: its memory goes, the window moves on
        P.consume_piece++;
        P.consume_batch = 0;
        P.cv.notify_all();
    }
}

bool SeqReader::next_batch(Batch &b, uint64_t max_bases, uint64_t max_reads, bool keep_ids) {
    if (par_) {
        if (!par_->started) {
            par_->max_bases = max_bases;
            par_->max_reads = max_reads;
        }
        return next_batch_parallel(b, keep_ids);
    }
    b.clear();
    b.first_record = n_records_;
    if (!err_.empty()) return false;
    for (;;) {
        if (b.bases.size() >= max_bases || b.n_reads() >= max_reads) return true;
        // header line (possibly carried over from the previous FASTA record)
        if (!have_pending_) {
            if (!read_line(pending_)) return false;  // clean EOF
            have_pending_ = true;
        }
        trim_end(pending_);
        if (pending_.empty()) {  // blank line between records
            have_pending_ = false;
            continue;
        }
        if (fmt_ == SeqFormat::Fasta) {
            if (pending_[0] != '>') {
                err_ = "Expected > at record start.";
                return false;
            }
            if (keep_ids) b.ids.push_back(first_token(pending_));
            have_pending_ = false;
            const char *lp;
            size_t ln;
            while (read_line_view(lp, ln)) {
                if (ln && lp[0] == '>') {
                    pending_.assign(lp, ln);
                    have_pending_ = true;
                    break;
                }
                trim_end_view(lp, ln);
                b.bases.insert(b.bases.end(), lp, lp + ln);
            }
        } else {
            if (pending_[0] != '@') {
                err_ = "Expected @ at record start.";
                return false;
            }
            if (keep_ids) b.ids.push_back(first_token(pending_));
            have_pending_ = false;
            uint64_t seq_lines = 0;
            const size_t start = b.bases.size();
            bool plus = false;
            const char *lp;
            size_t ln;
            while (read_line_view(lp, ln)) {
                if (ln && lp[0] == '+') {
                    plus = true;
                    break;
                }
                trim_end_view(lp, ln);
                b.bases.insert(b.bases.end(), lp, lp + ln);
                seq_lines++;
            }
            if (!plus) {
                err_ = "Incomplete record. Each FastQ record has to consist of 4 lines: header, sequence, separator and qualities.";
                return false;
            }
            uint64_t qual = 0;
            for (uint64_t i = 0; i < seq_lines; i++) {
                if (!read_line_view(lp, ln)) break;
                trim_end_view(lp, ln);
                qual += ln;
            }
            if (qual != b.bases.size() - start) {
                err_ = "Unequal length of sequence an qualities.";
                return false;
            }
        }
        b.offsets.push_back(b.bases.size());
        n_records_++;
    }
}

bool SeqReader::seq_stats(const std::string &path, uint64_t &seq_count, uint64_t &total_length, std::string &err) {
    SeqReader r;
    seq_count = 0;
    total_length = 0;
    if (!r.open(path, false)) {
        err = r.error();
        return false;
    }
    Batch b;
    for (;;) {
        const bool more = r.next_batch(b, 64ull << 20, 1ull << 22);
        seq_count += b.n_reads();
        total_length += b.bases.size();
        if (!more) break;
    }
    if (r.failed()) {
        err = r.error();
        return false;
    }
    return true;
}

}  // namespace kthost
