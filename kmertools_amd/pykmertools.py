"""Mirror of the reference's `pykmertools` Python module (pybindings/src/{oligo,kmer}.rs,
pip/src/lib.rs:31-40), running on the MI355X through libkmertools_hip.so.

Same class names, argument names, defaults and error behaviour:

    OligoComputer(ksize).vectorise_one(seq, norm=True, mins=True)      pybindings/src/oligo.rs:22,39
                        .vectorise_batch(seqs, norm=True, mins=True)   :77
                        .get_header(mins=True)                         :85
    CgrComputer(vecsize).vectorise_one(seq) / .vectorise_batch(seqs)   pybindings/src/cgr.rs:22,38,60
    MinimiserGenerator(seq, wsize, msize)  iterable of (mmer, start, end), .to_acgt(mmer)   pybindings/src/min.rs:24,36,46
    KmerGenerator(seq, ksize)  iterable of (fwd, rev)                  pybindings/src/kmer.rs:22,39
                 .kmer_pos_maps()                                      :31
    utils.to_acgt(kmer, ksize) / utils.to_numeric(kmer)                :49,:57

Quirks kept on purpose: raw mode (`mins=False`) normalises by
2x the k-mer count (pybindings/src/oligo.rs:61); `kmer_pos_maps()` returns
(list[4^k], dict pos->kmer, count); `to_numeric` raises ValueError beyond 32 bases.
"""
import numpy as np

from . import _lib, device

_ctx = None


def _context():
    """process-wide Context on device 0 with a private stream (host-array calling style)"""
    global _ctx
    if _ctx is None:
        _ctx = device.Context(0)
    return _ctx


class OligoComputer:
    """Computer for generating oligonucleotide frequency vectors"""

    def __init__(self, ksize):
        self.ksize = int(ksize)
        # the reference takes any usize here (pybindings/src/oligo.rs:22-31; only the CLI restricts k to 3..=7,
        # kmertools/src/args.rs:85) and allocates a 4^k-entry rank map: this build goes as far as k = 12 (k = 3..7 on
        # the LDS kernel, the others counted in global memory)
        if not 1 <= self.ksize <= 12:
            raise ValueError("ksize must be in 1..=12")
        self.kcount = device.bins(self.ksize, True)

    def vectorise_one(self, seq, norm=True, mins=True):
        return self.vectorise_batch([seq], norm, mins)[0]

    def vectorise_batch(self, seqs, norm=True, mins=True):
        return self.vectorise_batch_numpy(seqs, norm, mins).tolist()

    def vectorise_batch_numpy(self, seqs, norm=True, mins=True):
        """extension: the same result as an (n, bins) float64 ndarray"""
        bases, offsets = device.to_csr(seqs)
        # CLI crate adds 1 per k-mer to `total`; the python binding adds 2 in raw mode
        step = 1 if mins else 2
        return _context().oligo_host(bases, offsets, self.ksize, count_min=mins, norm=norm, total_step=step)

    def get_header(self, mins=True):
        if mins:
            _, pos_kmer, _ = device.pos_map(self.ksize)
            return [device.numeric_to_kmer(int(x), self.ksize) for x in pos_kmer]
        return [device.numeric_to_kmer(x, self.ksize) for x in range(4 ** self.ksize)]


class CgrComputer:
    """Computer for generating chaos game representation (cgr)"""

    def __init__(self, vecsize):
        self.vecsize = int(vecsize)
        if self.vecsize < 0:
            raise OverflowError("can't convert negative int to unsigned")  # pyo3's usize extraction

    def vectorise_one(self, seq):
        return self.vectorise_batch([seq])[0]

    def vectorise_batch(self, seqs):
        """list of sequences -> list of lists of (x, y) tuples; ValueError("Bad nucleotide, unable to
        proceed") if any sequence holds a byte outside ACGTUacgtu (pybindings/src/cgr.rs:47-49)"""
        bases, offsets = device.to_csr(seqs)
        try:
            xy = _context().cgr_host(bases, offsets, self.vecsize)
        except _lib.KmertoolsError as e:
            if e.code == _lib.KT_ERR_BADNT:
                raise ValueError("Bad nucleotide, unable to proceed") from None
            raise
        pts = list(map(tuple, xy.tolist()))
        return [pts[int(offsets[i]):int(offsets[i + 1])] for i in range(len(offsets) - 1)]


class MinimiserGenerator:
    """An iterator object to iterate minimisers as (kmer, start, end) numeric minimiser tuples"""

    def __init__(self, seq, wsize, msize):
        self.msize = int(msize)
        wsize = int(wsize)
        if not 1 <= self.msize <= 31:
            raise ValueError("msize must be in 1..31")
        if wsize < self.msize:
            # the reference computes `wsize - msize + 1` in usize here (kmer/src/minimiser.rs:53) and panics
            raise ValueError("wsize must not be smaller than msize")
        bases, offsets = device.to_csr([seq])
        evo, k, s, e = _context().minimisers_host(bases, offsets, wsize, self.msize)
        self._items = list(zip(k.tolist(), s.tolist(), e.tolist()))
        self._i = 0

    def to_acgt(self, mmer):
        return device.numeric_to_kmer(int(mmer), self.msize)

    def __iter__(self):
        return self

    def __next__(self):
        if self._i >= len(self._items):
            raise StopIteration
        self._i += 1
        return self._items[self._i - 1]


class KmerGenerator:
    """Computer for generating k-mers: iterates (forward, reverse-complement) pairs"""

    def __init__(self, seq, ksize):
        self.ksize = int(ksize)
        if not 1 <= self.ksize <= 31:
            raise ValueError("ksize must be in 1..=31")
        bases, offsets = device.to_csr([seq])
        fwd, rev, _ = _context().kmers_host(bases, offsets, self.ksize)
        self._it = iter(zip(fwd.tolist(), rev.tolist()))

    def kmer_pos_maps(self):
        m, pos_kmer, count = device.pos_map(self.ksize)
        return m.tolist(), {i: int(x) for i, x in enumerate(pos_kmer)}, count

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._it)


class _Utils:
    @staticmethod
    def to_acgt(kmer, ksize):
        """Translate numeric k-mer to ACGT"""
        return device.numeric_to_kmer(int(kmer), int(ksize))

    @staticmethod
    def to_numeric(kmer):
        """Translate ACGT kmer to numeric pair"""
        if len(kmer) > 32:
            raise ValueError("Invalid k-mer length: %d, must be <= 32" % len(kmer))
        return device.kmer_to_numeric(kmer)


utils = _Utils()


def run_cli(argv=None):
    """The `kmertools` command line with this process's arguments (pip/src/lib.rs:11-18: `run_cli` parses
    std::env::args_os().skip(1) and runs the CLI).  Here the CLI is the C++ binary built next to the library; it is
    started as a fresh child process (never exec'ed from a process that may have initialised the GPU) and its exit
    status is returned."""
    import subprocess
    import sys
    exe = _lib._HERE / "bin" / "kmertools"
    if not exe.exists():
        raise FileNotFoundError("%s is missing - build it with `make -C kmertools_amd/csrc`" % exe)
    args = list(sys.argv[1:] if argv is None else argv)
    return subprocess.call([str(exe)] + args)
