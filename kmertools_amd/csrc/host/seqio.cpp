#include "seqio.hpp"

#include <ctype.h>
#include <string.h>

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

SeqReader::~SeqReader() {
    if (gz_) gzclose(gz_);
}

bool SeqReader::open(const std::string &path, bool sniff) {
    err_.clear();
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

bool SeqReader::fill() {
    if (eof_) return false;
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
    pos_ = 0;
    end_ = (size_t)n;
    return true;
}

bool SeqReader::peek(int &c) {
    if (pos_ == end_ && !fill()) return false;
    c = buf_[pos_];
    return true;
}

bool SeqReader::read_line(std::string &line) {
    line.clear();
    bool any = false;
    for (;;) {
        if (pos_ == end_ && !fill()) return any;
        any = true;
        const unsigned char *p = buf_.data() + pos_;
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
    const unsigned char *b = buf_.data() + pos_;
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

bool SeqReader::next_batch(Batch &b, uint64_t max_bases, uint64_t max_reads, bool keep_ids) {
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
