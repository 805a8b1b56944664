"""ctypes wrapper around oracle/libkt_oracle.so - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It is the checker, never the thing measured or shipped; nothing in
kmertools_amd/ imports it.

Also holds the text formatters that restate the reference's output formats
(composition/src/oligo.rs:125-145,201-217; composition/src/oligocgr.rs:93-97;
counter/src/lib.rs:220-230) so fixture files can be compared byte-for-byte.
"""
import ctypes as C
import os
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_LIB = None

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)


def build(force=False):
    so = _HERE / "libkt_oracle.so"
    src = _HERE / "kt_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "libkt_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = C.CDLL(str(so))
        L.kto_nt4.restype = C.c_uint8
        L.kto_nt4.argtypes = [C.c_uint8]
        L.kto_kmers.restype = C.c_uint64
        L.kto_kmers.argtypes = [u8p, C.c_uint64, C.c_uint64, u64p, u64p, u64p]
        L.kto_rev_comp.restype = C.c_uint64
        L.kto_rev_comp.argtypes = [C.c_uint64, C.c_uint64]
        L.kto_pos_maps.restype = C.c_uint64
        L.kto_pos_maps.argtypes = [C.c_uint64, u64p, u64p]
        L.kto_numeric_to_kmer.restype = None
        L.kto_numeric_to_kmer.argtypes = [C.c_uint64, C.c_uint64, C.c_char_p]
        L.kto_kmer_to_numeric.restype = None
        L.kto_kmer_to_numeric.argtypes = [C.c_char_p, C.c_uint64, u64p, u64p]
        L.kto_oligo_one.restype = None
        L.kto_oligo_one.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_double,
                                    u64p, C.c_uint64, f64p]
        L.kto_oligo_batch.restype = C.c_int
        L.kto_oligo_batch.argtypes = [u8p, u64p, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                                      C.c_double, f64p, C.c_int]
        L.kto_cgr_coords.restype = None
        L.kto_cgr_coords.argtypes = [C.c_uint64, C.c_double, f64p]
        L.kto_counter_new.restype = C.c_void_p
        L.kto_counter_new.argtypes = [C.c_uint64]
        L.kto_counter_free.restype = None
        L.kto_counter_free.argtypes = [C.c_void_p]
        L.kto_counter_add_reads.restype = C.c_int
        L.kto_counter_add_reads.argtypes = [C.c_void_p, u8p, u64p, C.c_uint64, C.c_uint64, C.c_int]
        L.kto_counter_reserve.restype = None
        L.kto_counter_reserve.argtypes = [C.c_void_p, C.c_uint64]
        L.kto_counter_add_pairs.restype = C.c_int
        L.kto_counter_add_pairs.argtypes = [C.c_void_p, u64p, u32p, C.c_uint64]
        L.kto_counter_size.restype = C.c_uint64
        L.kto_counter_size.argtypes = [C.c_void_p]
        L.kto_counter_export.restype = C.c_uint64
        L.kto_counter_export.argtypes = [C.c_void_p, u64p, u32p, C.c_int]
        L.kto_counter_spill.restype = C.c_int
        L.kto_counter_spill.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int]
        L.kto_merge.restype = C.c_uint64
        L.kto_merge.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_int]
        L.kto_cgr_one.restype = C.c_uint64
        L.kto_cgr_one.argtypes = [u8p, C.c_uint64, C.c_double, f64p]
        L.kto_minimisers.restype = C.c_uint64
        L.kto_minimisers.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint64, u64p, u64p, u64p]
        L.kto_minimisers_batch.restype = C.c_uint64
        L.kto_minimisers_batch.argtypes = [u8p, u64p, C.c_uint64, C.c_uint64, C.c_uint64, u64p, u64p, u64p, C.c_uint64]
        L.kto_cgr_batch.restype = C.c_uint64
        L.kto_cgr_batch.argtypes = [u8p, u64p, C.c_uint64, C.c_double, f64p]
        L.kto_cov_batch.restype = C.c_int
        L.kto_cov_batch.argtypes = [C.c_void_p, u8p, u64p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, f64p]
        L.kto_matrix_text_file.restype = C.c_int
        L.kto_matrix_text_file.argtypes = [f64p, C.c_uint64, C.c_uint64, C.c_int, C.c_char, f64p, C.c_char_p, C.c_int,
                                           C.c_int]
        L.kto_fmt_display.restype = C.c_int
        L.kto_fmt_display.argtypes = [C.c_double, C.c_char_p]
        L.kto_synth_reads.restype = None
        L.kto_synth_reads.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int,
                                      C.c_uint64, u8p]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def to_csr(seqs):
    """list of bytes/str -> (bases u8[total], offsets u64[n+1])"""
    bs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    bases = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    if bases.size == 0:
        bases = np.zeros(1, np.uint8)[:0]
    return bases, offsets


def nt4(c):
    return int(lib().kto_nt4(c))


def kmers(seq, k):
    """[(fwd, rev)] in positional order + end positions (kmer.rs:80-106)."""
    b = np.frombuffer(seq.encode("latin-1") if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
    n = len(b)
    buf = np.zeros(max(n, 1), dtype=np.uint8)
    buf[:n] = b
    fwd = np.zeros(max(n, 1), np.uint64)
    rev = np.zeros(max(n, 1), np.uint64)
    end = np.zeros(max(n, 1), np.uint64)
    c = lib().kto_kmers(_p(buf, u8p), n, k, _p(fwd, u64p), _p(rev, u64p), _p(end, u64p))
    return fwd[:c].copy(), rev[:c].copy(), end[:c].copy()


def rev_comp(kmer, k):
    return int(lib().kto_rev_comp(kmer, k))


def pos_maps(k):
    """(min_mer_pos_map[4^k], pos_min_mer[count], count)  kmer.rs:54-73"""
    n = 4 ** k
    m = np.zeros(n, np.uint64)
    pk = np.zeros(n, np.uint64)
    c = lib().kto_pos_maps(k, _p(m, u64p), _p(pk, u64p))
    return m, pk[:c].copy(), int(c)


def numeric_to_kmer(kmer, k):
    buf = C.create_string_buffer(k + 1)
    lib().kto_numeric_to_kmer(kmer, k, buf)
    return buf.value.decode()


def kmer_to_numeric(s):
    f = C.c_uint64()
    r = C.c_uint64()
    b = s.encode("latin-1")
    lib().kto_kmer_to_numeric(b, len(b), C.byref(f), C.byref(r))
    return f.value, r.value


def bins_for(k, count_min):
    return pos_maps(k)[2] if count_min else 4 ** k


def oligo_batch(bases, offsets, k, count_min=True, norm=True, total_step=1.0, threads=1):
    n = len(offsets) - 1
    bins = bins_for(k, count_min)
    out = np.zeros((n, bins), dtype=np.float64)
    bb = bases if bases.size else np.zeros(1, np.uint8)
    lib().kto_oligo_batch(_p(bb, u8p), _p(offsets, u64p), n, k, int(count_min), int(norm),
                          float(total_step), _p(out, f64p), threads)
    return out


def oligo_one(seq, k, count_min=True, norm=True, total_step=1.0):
    bases, offsets = to_csr([seq])
    return oligo_batch(bases, offsets, k, count_min, norm, total_step)[0]


def header(k, count_min=True):
    """oligo.rs:69-83 get_header"""
    if count_min:
        _, pk, _ = pos_maps(k)
        return [numeric_to_kmer(int(x), k) for x in pk]
    return [numeric_to_kmer(x, k) for x in range(4 ** k)]


def cgr_coords(k, vecsize):
    c = pos_maps(k)[2]
    xy = np.zeros((c, 2), dtype=np.float64)
    lib().kto_cgr_coords(k, float(vecsize), _p(xy, f64p))
    return xy


def minimisers(seq, wsize, msize):
    """kmer/src/minimiser.rs:61-175: [(minimiser, window_start, window_end)]; wsize 0 = whole sequence
    (misc/src/minimisers.rs:44-48)"""
    b = np.frombuffer(seq.encode("latin-1") if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
    n = len(b)
    if wsize == 0:
        wsize = n
    buf = np.zeros(max(n, 1), np.uint8)
    buf[:n] = b
    k = np.zeros(n + 2, np.uint64)
    s = np.zeros(n + 2, np.uint64)
    e = np.zeros(n + 2, np.uint64)
    c = lib().kto_minimisers(_p(buf, u8p), n, wsize, msize, _p(k, u64p), _p(s, u64p), _p(e, u64p))
    return [(int(k[i]), int(s[i]), int(e[i])) for i in range(c)]


def minimisers_batch_count(bases, offsets, wsize, msize, cap=1 << 22):
    """CPU-baseline helper: runs the iterator over every read (< 4 kb) of a CSR batch, returns the triple count"""
    k = np.zeros(cap, np.uint64)
    s = np.zeros(cap, np.uint64)
    e = np.zeros(cap, np.uint64)
    return int(lib().kto_minimisers_batch(_p(bases, u8p), _p(offsets, u64p), len(offsets) - 1, wsize, msize,
                                          _p(k, u64p), _p(s, u64p), _p(e, u64p), cap))


def seq_to_min_lines(records, wsize, msize):
    """misc/src/minimisers.rs:131-141: "id\tKMER:s-e\t...\t\n" per record"""
    out = []
    for rid, seq in records:
        parts = [rid] + ["%s:%d-%d" % (numeric_to_kmer(k, msize), s, e) for k, s, e in minimisers(seq, wsize, msize)]
        parts.append("\n")
        out.append("\t".join(parts))
    return out


def bin_sequences_lines(records, wsize, msize):
    """misc/src/minimisers.rs:49-54,81-84: minimiser -> [(id, start, end)] in Rust Debug format"""
    table = {}
    for rid, seq in records:
        for k, s, e in minimisers(seq, wsize, msize):
            table.setdefault(numeric_to_kmer(k, msize), []).append((rid, s, e))
    return ["%s\t[%s]\n" % (k, ", ".join('("%s", %d, %d)' % t for t in v)) for k, v in table.items()]


def cgr_points(seq, vecsize=1):
    """composition/src/cgr.rs:127-144: (n, 2) f64 points; ValueError on a byte outside ACGTUacgtu"""
    b = np.frombuffer(seq.encode("latin-1") if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
    n = len(b)
    buf = np.zeros(max(n, 1), np.uint8)
    buf[:n] = b
    out = np.zeros((max(n, 1), 2), np.float64)
    bad = lib().kto_cgr_one(_p(buf, u8p), n, float(vecsize), _p(out, f64p))
    if bad:
        raise ValueError("Bad nucleotide, unable to proceed")
    return out[:n].copy()


def cgr_batch(bases, offsets, vecsize=1, out=None):
    """all reads of a CSR batch -> (total_bases, 2) f64 (`out`: a preallocated result to fill)"""
    total = int(offsets[-1])
    if out is None:
        out = np.zeros((max(total, 1), 2), np.float64)
    bb = bases if bases.size else np.zeros(1, np.uint8)
    if lib().kto_cgr_batch(_p(bb, u8p), _p(offsets, u64p), len(offsets) - 1, float(vecsize), _p(out, f64p)):
        raise ValueError("Bad nucleotide, unable to proceed")
    return out[:total]


def cgr_text(point_rows):
    """composition/src/cgr.rs:97-104: "(x,y)" joined by " ", one line per read"""
    lines = []
    for pts in point_rows:
        lines.append(" ".join("(%s,%s)" % (fmt_display(float(x)), fmt_display(float(y))) for x, y in pts) + "\n")
    return "".join(lines).encode()


class Counter:
    """In-memory restatement of counter/src/lib.rs count_chunk + merge."""

    def __init__(self, n_parts=1):
        self.h = lib().kto_counter_new(n_parts)

    def add_reads(self, bases, offsets, k, threads=1):
        bb = bases if bases.size else np.zeros(1, np.uint8)
        lib().kto_counter_add_reads(self.h, _p(bb, u8p), _p(offsets, u64p), len(offsets) - 1, k, threads)

    def reserve(self, keys):
        """room for `keys` distinct k-mers (scc grows while it adds; this map is sized between threaded phases)"""
        lib().kto_counter_reserve(self.h, int(keys))

    def add_pairs(self, keys, counts):
        keys = np.ascontiguousarray(keys, np.uint64)
        counts = np.ascontiguousarray(counts, np.uint32)
        if len(keys):
            lib().kto_counter_add_pairs(self.h, _p(keys, u64p), _p(counts, u32p), len(keys))

    def size(self):
        return int(lib().kto_counter_size(self.h))

    def spill(self, out_dir, chunk, threads=1):
        """count_chunk's text spill (counter/src/lib.rs:151-167): temp_kmers.part_P_chunk_<chunk> per partition"""
        if lib().kto_counter_spill(self.h, str(out_dir).encode(), chunk, threads):
            raise OSError("spill to %s failed" % out_dir)

    def cov_batch(self, bases, offsets, k, bin_size, bin_count, norm=True):
        """coverage/src/lib.rs:165-184 over a CSR batch -> (n_reads, bin_count) f64"""
        n = len(offsets) - 1
        out = np.zeros((n, bin_count), np.float64)
        bb = bases if bases.size else np.zeros(1, np.uint8)
        lib().kto_cov_batch(self.h, _p(bb, u8p), _p(offsets, u64p), n, k, bin_size, bin_count, int(norm), _p(out, f64p))
        return out

    def export(self, sorted_=True):
        n = self.size()
        keys = np.zeros(max(n, 1), np.uint64)
        counts = np.zeros(max(n, 1), np.uint32)
        lib().kto_counter_export(self.h, _p(keys, u64p), _p(counts, u32p), int(sorted_))
        return keys[:n].copy(), counts[:n].copy()

    def __del__(self):
        if getattr(self, "h", None) and _LIB is not None:
            _LIB.kto_counter_free(self.h)
            self.h = None


def merge_files(out_dir, n_parts, chunks, threads=1, acgt=False, k=0, delete=True):
    """CountComputer::merge (counter/src/lib.rs:172-234) over the chunk files in out_dir -> lines of kmers.counts"""
    n = lib().kto_merge(str(out_dir).encode(), n_parts, chunks, threads, int(acgt), k, int(delete))
    if n == 2 ** 64 - 1:
        raise OSError("merge in %s failed" % out_dir)
    return int(n)


def count_reads(bases, offsets, k, n_parts=1, threads=1):
    c = Counter(n_parts)
    c.add_reads(bases, offsets, k, threads)
    return c.export(True)


def synth_reads(seed, n_reads, read_len, noise=False, genome_len=0, first_read=0):
    bases = np.zeros(max(n_reads * read_len, 1), np.uint8)
    lib().kto_synth_reads(seed, first_read, n_reads, read_len, int(noise), genome_len, _p(bases, u8p))
    bases = bases[: n_reads * read_len]
    offsets = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)).astype(np.uint64)
    return bases, offsets


# --------------------------------------------------------------------------
# Text formats of the reference (restated)

def fmt_fixed6(x):
    """Rust `format!("{:.6}", x)` (oligo.rs:132-134, NUMBER_SIZE-2 = 6)."""
    return "%.6f" % x


def fmt_display(x):
    """Rust `Display` for f64: shortest round-trip digits, positional notation only,
    integral values without a fractional part (oligo.rs:136, oligocgr.rs:95)."""
    x = float(x)
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    r = repr(x)
    if "e" in r or "E" in r:
        mant, exp = r.lower().split("e")
        exp = int(exp)
        sign = ""
        if mant.startswith("-"):
            sign, mant = "-", mant[1:]
        if "." in mant:
            ip, fp = mant.split(".")
        else:
            ip, fp = mant, ""
        digits = ip + fp
        point = len(ip) + exp
        if point <= 0:
            r = sign + "0." + "0" * (-point) + digits
        elif point >= len(digits):
            r = sign + digits + "0" * (point - len(digits))
        else:
            r = sign + digits[:point] + "." + digits[point:]
        r = r.rstrip("0").rstrip(".") if "." in r else r
    if r.endswith(".0"):
        r = r[:-2]
    return r


def oligo_text(mat, norm, delim=" ", header_line=None):
    """Bytes of the `comp oligo` output file (oligo.rs:125-145 / :201-217)."""
    f = fmt_fixed6 if norm else fmt_display
    lines = []
    if header_line is not None:
        lines.append(delim.join(header_line) + "\n")
    for row in mat:
        lines.append(delim.join(f(v) for v in row) + "\n")
    return "".join(lines).encode()


def matrix_text_file(mat, path, mode="fixed6", delim=" ", xy=None, append=False, threads=8):
    """Writes / appends the reference's text of a row matrix to `path` with the C formatter (multi-batch CLI
    comparisons: too many numbers for the Python formatters above, which it is tested against).
    mode: "fixed6" ({:.6}), "display" (Rust Display) or "cgr" ("(x,y,v)" triples, xy = cgr_coords)."""
    mat = np.ascontiguousarray(mat, np.float64)
    m = {"display": 0, "fixed6": 1, "cgr": 2}[mode]
    xyp = None if xy is None else _p(np.ascontiguousarray(xy, np.float64), f64p)
    rc = lib().kto_matrix_text_file(_p(mat, f64p), mat.shape[0], mat.shape[1], m, delim.encode(), xyp,
                                    str(path).encode(), int(append), threads)
    if rc:
        raise OSError("kto_matrix_text_file(%s) failed" % path)


def fmt_display_c(x):
    buf = C.create_string_buffer(400)
    lib().kto_fmt_display(float(x), buf)
    return buf.value.decode()


def oligocgr_text(mat, xy):
    """Bytes of `comp cgr -k` output (oligocgr.rs:93-97)."""
    lines = []
    for row in mat:
        lines.append(" ".join("(%s,%s,%s)" % (fmt_display(xy[i, 0]), fmt_display(xy[i, 1]), fmt_display(v))
                              for i, v in enumerate(row)) + "\n")
    return "".join(lines).encode()


def counts_lines(keys, counts, k=None, acgt=False):
    """Sorted lines of kmers.counts (counter/src/lib.rs:220-230; tests sort, :271-275)."""
    if acgt:
        return sorted("%s\t%d" % (numeric_to_kmer(int(a), k), int(b)) for a, b in zip(keys, counts))
    return sorted("%d\t%d" % (int(a), int(b)) for a, b in zip(keys, counts))


# --------------------------------------------------------------------------
# Minimal FASTA/FASTQ parsing for fixtures (the reference uses bio 2.3.0,
# ktio/src/seq.rs:100-131: id = first header token, sequence lines joined).

def read_records(path):
    import gzip
    p = str(path)
    op = gzip.open if p.endswith(".gz") else open
    with op(p, "rb") as fh:
        data = fh.read()
    lines = data.split(b"\n")
    recs = []
    if data[:1] == b">":
        cur_id, cur = None, []
        for ln in lines:
            ln = ln.rstrip(b"\r")
            if ln.startswith(b">"):
                if cur_id is not None:
                    recs.append((cur_id, b"".join(cur)))
                cur_id = ln[1:].split()[0].decode() if ln[1:].split() else ""
                cur = []
            elif cur_id is not None:
                cur.append(ln)
        if cur_id is not None:
            recs.append((cur_id, b"".join(cur)))
    else:
        i = 0
        while i + 3 < len(lines) + 1 and i < len(lines) and lines[i].startswith(b"@"):
            rid = lines[i][1:].split()[0].decode()
            recs.append((rid, lines[i + 1].rstrip(b"\r")))
            i += 4
    return recs
