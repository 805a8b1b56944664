"""A numpy restatement of what the sharded counter puts on the wires (kmertools_amd/csrc/kt_superkmer.hpp, kt_shard.hip's
route_kernel) - test infrastructure, written from the header's prose, not from the kernel: the owner of a k-mer by its
minimiser, the cut of a read into records of at most 8 k-mers of one owner, the 80-bit record, the blocks of 1024.
The reference has no counterpart (its partition is `min_mer % n_parts`, counter/src/lib.rs:127 - not observable in its
output); what these helpers pin is that the library's host function, its device kernel and this file agree."""
import numpy as np

REC_KMERS = 8
BLOCK_RECS = 1024
BLOCK_WORDS = 1280
M32 = np.uint64(0xFFFFFFFF)


def window_of(k):
    w = 16
    while w > 1 and k < w + 7:
        w >>= 1
    return w


def mmer_of(k):
    return k - window_of(k) + 1


def _mul24(a, b):
    return ((a & np.uint64(0xFFFFFF)) * (np.uint64(b) & np.uint64(0xFFFFFF))) & M32


def _mhash(x):
    x = x.astype(np.uint64) & M32
    return (_mul24(x, 0x9E3779) + _mul24(x >> np.uint64(12), 0x85EBCB)) & M32


def _owner_of_min(h, n):
    x = (_mul24(h, 0x9E3779) + _mul24(h >> np.uint64(8), 0x85EBCB)) & M32
    x ^= x >> np.uint64(16)
    return (_mul24(x >> np.uint64(8), n) >> np.uint64(24)).astype(np.int64)


def _rev_comp(x, m):
    """reverse complement of m-mers held in the low 2m bits"""
    x = x.astype(np.uint64)
    out = np.zeros_like(x)
    for j in range(m):
        out |= (np.uint64(3) - ((x >> np.uint64(2 * j)) & np.uint64(3))) << np.uint64(2 * (m - 1 - j))
    return out


def owner_of_kmers(kmers, k, n):
    """owners of k-mers (either strand; u64 array) among n ranks"""
    kmers = np.asarray(kmers, dtype=np.uint64)
    if n <= 1:
        return np.zeros(len(kmers), np.int64)
    m, w = mmer_of(k), window_of(k)
    mask = np.uint64((1 << (2 * m)) - 1)
    best = np.full(len(kmers), 0xFFFFFFFF, np.uint64)
    for i in range(w):
        f = (kmers >> np.uint64(2 * (k - m - i))) & mask
        best = np.minimum(best, _mhash(np.minimum(f, _rev_comp(f, m))))
    return _owner_of_min(best, n)


def records_of_read(codes, valid, k, n):
    """codes: 2-bit codes of a read (u8 array), valid: bool per base -> list of (owner, start, n_kmers): the records, in
    read order - maximal runs of consecutive k-mers with one owner, cut every 8 k-mers from the run's start"""
    L = len(codes)
    if L < k:
        return []
    nk = L - k + 1
    ok = np.ones(nk, bool)
    bad = np.flatnonzero(~valid)
    for b in bad:
        ok[max(0, b - k + 1):min(nk, b + 1)] = False
    fwd = np.zeros(nk, np.uint64)
    for j in range(k):
        fwd = (fwd << np.uint64(2)) | codes[j:j + nk].astype(np.uint64)
    own = owner_of_kmers(fwd, k, n)
    recs = []
    s = 0
    while s < nk:
        if not ok[s]:
            s += 1
            continue
        e = s
        while e + 1 < nk and ok[e + 1] and own[e + 1] == own[s]:
            e += 1
        for r in range(s, e + 1, REC_KMERS):
            recs.append((int(own[s]), r, min(REC_KMERS, e + 1 - r)))
        s = e + 1
    return recs


def pack_record(codes, start, n_kmers, k):
    """-> (a u64, b u16) of the record that holds k-mers start .. start + n_kmers - 1 of the read"""
    nb = n_kmers + k - 1
    a = 0
    bb = 0
    for j in range(nb):
        c = int(codes[start + j])
        if j < 32:
            a |= c << (62 - 2 * j)
        else:
            bb |= c << (10 - 2 * (j - 32))
    return a, (bb << 4) | n_kmers


def kmers_of_record(a, b, k):
    """canonical k-mers of a record (python ints)"""
    n = b & 15
    bases = [(a >> (62 - 2 * j)) & 3 for j in range(32)] + [((b >> 4) >> (10 - 2 * j)) & 3 for j in range(6)]
    out = []
    for s in range(n):
        f = 0
        r = 0
        for j in range(k):
            f = (f << 2) | bases[s + j]
            r |= (3 - bases[s + j]) << (2 * j)
        out.append(min(f, r))
    return out


def blocks_of_records(recs):
    """[(a, b)] -> u64 array of whole blocks (1024 a, then 1024 b as u16)"""
    nb = (len(recs) + BLOCK_RECS - 1) // BLOCK_RECS
    out = np.zeros(nb * BLOCK_WORDS, np.uint64)
    for i, (a, b) in enumerate(recs):
        blk, idx = divmod(i, BLOCK_RECS)
        out[blk * BLOCK_WORDS + idx] = np.uint64(a)
        out[blk * BLOCK_WORDS + BLOCK_RECS:(blk + 1) * BLOCK_WORDS].view(np.uint16)[idx] = b
    return out


def records_of_blocks(words, n_rec):
    out = []
    for i in range(n_rec):
        blk, idx = divmod(i, BLOCK_RECS)
        a = int(words[blk * BLOCK_WORDS + idx])
        b = int(words[blk * BLOCK_WORDS + BLOCK_RECS:(blk + 1) * BLOCK_WORDS].view(np.uint16)[idx])
        out.append((a, b))
    return out


_NT = np.full(256, 4, np.uint8)
for _c, _v in ((b"Aa", 0), (b"Cc", 1), (b"Gg", 2), (b"TtUu", 3)):
    for _x in _c:
        _NT[_x] = _v
for _v in range(4):
    _NT[_v] = _v


def codes_of(seq_bytes):
    """ASCII read -> (2-bit codes u8, valid bool), the reference's alphabet (kmer/src/kmer.rs:6-15)"""
    e = _NT[np.frombuffer(bytes(seq_bytes), np.uint8)]
    return (e & 3).astype(np.uint8), e < 4
