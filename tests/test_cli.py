"""The C++ host side (kmertools_amd/csrc/host): reader behaviour and argument handling on CPU,
and - with a GPU - byte-for-byte equality of the CLI's output files with the reference's
golden files (the reference's own end-to-end tests: composition/src/oligo.rs:312-432,
composition/src/oligocgr.rs:223-238, counter/src/lib.rs:260-276)."""
import gzip
import pathlib
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
CLI = ROOT / "kmertools_amd" / "bin" / "kmertools"


@pytest.fixture(scope="module")
def cli():
    if not CLI.exists():
        subprocess.check_call(["make", "-C", str(ROOT / "kmertools_amd" / "csrc"), "-j4"], stdout=subprocess.DEVNULL)
    return str(CLI)


def run(cli, *args, **kw):
    return subprocess.run([cli, *map(str, args)], capture_output=True, text=True, timeout=300, **kw)


def records(cli, path, *extra):
    r = run(cli, "debug-read", path, *extra)
    assert r.returncode == 0, r.stderr
    recs, meta = [], {}
    for ln in r.stdout.splitlines():
        parts = ln.split("\t")
        if ln.startswith("#"):
            meta[parts[0]] = (int(parts[1]), int(parts[2]))
        else:
            recs.append((int(parts[0]), parts[1], parts[2] if len(parts) > 2 else ""))
    return recs, meta


def test_reader_fixtures(cli, golden, kat):
    e = kat["reader"]
    for name, ids in [("reads.fq", e["fq_ids"]), ("reads.fa", e["fa_ids"]), ("reads.fq.gz", e["fq_ids"])]:
        recs, meta = records(cli, golden / name)
        assert [r[1] for r in recs] == ids and [r[2] for r in recs] == e["seqs"]
        assert [r[0] for r in recs] == [0, 1]                      # Sequence::n ordinals
        assert meta["#records"] == (2, e["total_length"]) == meta["#seq_stats"]


def test_reader_edge_cases(cli, tmp_path):
    # multi-line FASTA/FASTQ, CRLF, blank lines, lower case preserved, descriptions dropped,
    # empty sequence, no trailing newline, format sniffing of an unknown extension
    fa = tmp_path / "x.fa"
    fa.write_bytes(b">r1 some description\r\nACGT\r\nacgtn\r\n\r\n>r2\n>r3\tdesc\nTT\nGG")
    recs, meta = records(cli, fa)
    assert [(r[1], r[2]) for r in recs] == [("r1", "ACGTacgtn"), ("r2", ""), ("r3", "TTGG")]
    assert meta["#records"] == (3, 13)
    fq = tmp_path / "y.fastq"
    fq.write_bytes(b"@q1 d\nACGT\nAC\n+\nIIII\nII\n@q2\nNN\n+q2\n##\n")
    recs, _ = records(cli, fq)
    assert [(r[1], r[2]) for r in recs] == [("q1", "ACGTAC"), ("q2", "NN")]
    unk = tmp_path / "z.txt"
    unk.write_bytes(b">a\nAC\n")
    assert [(r[1], r[2]) for r in records(cli, unk)[0]] == [("a", "AC")]
    big = tmp_path / "big.fa.gz"
    seqs = ["ACGT" * (50 + i) for i in range(300)]
    with gzip.open(big, "wb") as fh:
        for i, s in enumerate(seqs):
            fh.write((">s%d\n" % i).encode())
            for j in range(0, len(s), 60):
                fh.write((s[j:j + 60] + "\n").encode())
    recs, meta = records(cli, big)
    assert [r[2] for r in recs] == seqs and meta["#seq_stats"] == (300, sum(map(len, seqs)))
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"@q\nACGT\n+\nII\n")
    r = run(cli, "debug-read", bad)
    assert r.returncode != 0 and "Unequal length" in r.stderr


def test_fixed6_formatter_matches_printf(cli):
    """the hand-written Rust-{:.6} formatter vs snprintf("%.6f") on every count ratio c/d (d <= 1200),
    the dyadic exact ties and a few hundred thousand random doubles"""
    r = run(cli, "debug-fixed6", "100000")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout


def test_cli_argument_handling(cli, golden, tmp_path):
    r = run(cli, "comp", "oligo", "-i", golden / "reads.fa", "-o", tmp_path / "o", "-k", "8")
    assert r.returncode == 2 and "is not in 3..=7" in r.stderr          # args.rs:85
    r = run(cli, "ctr", "-i", golden / "reads.fq", "-o", tmp_path / "c")
    assert r.returncode == 2 and "--k-size" in r.stderr                   # required, args.rs:219
    r = run(cli, "ctr", "-i", golden / "reads.fq", "-o", tmp_path / "c", "-k", "9")
    assert r.returncode == 2 and "10..=31" in r.stderr
    r = run(cli, "ctr", "-i", golden / "reads.fq", "-o", tmp_path / "c", "-k", "21", "-m", "3")
    assert r.returncode == 2 and "6..=128" in r.stderr
    r = run(cli, "comp", "oligo", "-i", golden / "reads.fa", "-o", tmp_path / "o", "-p", "xml")
    assert r.returncode == 2
    r = run(cli, "comp", "cgr", "-i", golden / "reads.fa", "-o", tmp_path / "o", "-c")
    assert "Error: cannot use counts in whole sequence CGR!" in r.stderr  # args.rs:284-287
    assert run(cli, "--help").returncode == 0 and run(cli, "comp", "oligo", "--help").returncode == 0
    assert run(cli, "cov").returncode == 2


def test_cli_fails_loudly_without_gpu(cli, golden, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = run(cli, "comp", "oligo", "-i", golden / "reads.fa", "-o", tmp_path / "o", "-k", "4")
    assert "Error: no HIP device" in r.stderr
    assert not (tmp_path / "o").exists() or (tmp_path / "o").stat().st_size == 0


@pytest.mark.gpu
def test_cli_golden_outputs(cli, golden, tmp_path):
    o = tmp_path / "fa.kmers"
    for inp in ("reads.fq", "reads.fa", "reads.fq.gz"):
        for t in (1, 8):
            r = run(cli, "comp", "oligo", "-i", golden / inp, "-o", o, "-k", "4", "-t", t)
            assert r.returncode == 0 and r.stderr == "", r.stderr
            assert o.read_bytes() == (golden / "expected_fa.kmers").read_bytes()
    run(cli, "comp", "oligo", "-i", golden / "reads.fq", "-o", o, "-k4", "--counts")
    assert o.read_bytes() == (golden / "expected_fa_batch_unnorm.kmers").read_bytes()
    run(cli, "comp", "oligo", "--input", golden / "reads.fq", "--output=%s" % o, "--k-size", "4", "-H")
    assert o.read_bytes() == (golden / "expected_fa_header.kmers").read_bytes()
    # stdin path (oligo.rs:89-91)
    with open(golden / "reads.fq", "rb") as fh:
        subprocess.run([cli, "comp", "oligo", "-i", "-", "-o", str(o), "-k", "4"], stdin=fh, check=True, timeout=300)
    assert o.read_bytes() == (golden / "expected_fa.kmers").read_bytes()
    # presets only change the delimiter
    run(cli, "comp", "oligo", "-i", golden / "reads.fq", "-o", o, "-k", "4", "-p", "csv")
    assert o.read_bytes() == (golden / "expected_fa.kmers").read_bytes().replace(b" ", b",")
    # raw mode: 256 columns
    run(cli, "comp", "oligo", "-i", golden / "reads.fq", "-o", o, "-k", "4", "-r", "-c")
    rows = o.read_text().splitlines()
    assert len(rows) == 2 and all(len(x.split()) == 256 for x in rows)
    assert sum(map(int, rows[0].split())) == 72 - 4 + 1
    # comp cgr -k 4 -c: default vecsize k^2 = 16 (args.rs:266-269)
    c = tmp_path / "k4.cgr"
    r = run(cli, "comp", "cgr", "-i", golden / "reads.fq", "-o", c, "-k", "4", "-c")
    assert r.returncode == 0 and r.stderr == ""
    assert c.read_bytes() == (golden / "expected_reads.k4.cgr").read_bytes()
    # ctr -k 15: sorted lines equal the reference's chunk file (all counts 1)
    d = tmp_path / "counts"
    r = run(cli, "ctr", "-i", golden / "reads.fq", "-o", d, "-k", "15")
    assert r.returncode == 0, r.stderr
    got = sorted((d / "kmers.counts").read_text().splitlines())
    assert got == sorted((golden / "expected_counts.part_0_chunk_0").read_text().splitlines())
    r = run(cli, "ctr", "-i", golden / "reads.fq", "-o", d, "-k", "15", "-a")
    lines = (d / "kmers.counts").read_text().splitlines()
    assert len(lines) == len(got) and all(len(x.split("\t")[0]) == 15 for x in lines)
    # whole-sequence CGR (no -k): vecsize defaults to 1 -> the reference's fixture (cgr.rs:189-199)
    w = tmp_path / "whole.cgr"
    r = run(cli, "comp", "cgr", "-i", golden / "reads.fq", "-o", w)
    assert r.returncode == 0 and r.stderr == ""
    assert w.read_bytes() == (golden / "expected_reads.cgr").read_bytes()
    r = run(cli, "comp", "cgr", "-i", golden / "reads.fq", "-o", w, "-c")
    assert r.returncode == 0 and "cannot use counts in whole sequence CGR" in r.stderr
    nfa = tmp_path / "n.fa"
    nfa.write_text(">a\nACGT\n>b\nACNT\n")
    r = run(cli, "comp", "cgr", "-i", nfa, "-o", w)
    assert r.returncode == 101 and "Bad nucleotide, unable to proceed" in r.stderr
    # min: the two presets against the reference's fixtures, compared like its own tests (trimmed, sorted lines)
    def trimmed(path):
        return sorted(ln.strip() for ln in path.read_text().splitlines())
    mo = tmp_path / "mins"
    r = run(cli, "min", "-i", golden / "reads.fq", "-o", mo, "-w", "31", "-m", "7")
    assert r.returncode == 0 and r.stderr == ""
    assert mo.read_text().endswith("\t\n") and trimmed(mo) == trimmed(golden / "expected_seq_minimisers")
    r = run(cli, "min", "-i", golden / "reads.fq", "-o", mo, "-p", "m2s")
    assert r.returncode == 0 and r.stderr == ""
    assert trimmed(mo) == trimmed(golden / "expected_minimisers")
    r = run(cli, "min", "-i", golden / "reads.fq", "-o", mo, "-w", "7", "-m", "7")
    assert r.returncode == 0 and "Window size must be longer than minimiser size!" in r.stderr
    r = run(cli, "min", "-i", golden / "reads.fq", "-o", mo, "-m", "29")
    assert r.returncode == 2 and "is not in 7..=28" in r.stderr
    # cov: the reference's test uses k=4 / bin_size 2 / bin_count 3 through the library API, which the
    # CLI's clap ranges (k 7..=31, bins >= 5) cannot express -> check the CLI against the oracle in
    # test_cli_larger_file_matches_oracle and the flag ranges here
    r = run(cli, "cov", "-i", golden / "reads.fq", "-o", tmp_path / "cv", "-k", "4")
    assert r.returncode == 2 and "is not in 7..=31" in r.stderr
    r = run(cli, "cov", "-i", golden / "reads.fq", "-o", tmp_path / "cv", "-s", "4")
    assert r.returncode == 2 and "--bin-size" in r.stderr
    # unwritable output -> the reference's message, exit code 0 (args.rs:260-262)
    r = run(cli, "comp", "oligo", "-i", golden / "reads.fq", "-o", tmp_path / "nodir" / "x", "-k", "4")
    assert r.returncode == 0 and "Error: Unable to write to file" in r.stderr
    r = run(cli, "comp", "oligo", "-i", tmp_path / "missing.fa", "-o", o, "-k", "4")
    assert "Error: Unable to open" in r.stderr


@pytest.mark.gpu
def test_cli_larger_file_matches_oracle(cli, oracle, tmp_path):
    """5000 ragged reads through file -> reader -> GPU -> text, vs the oracle's text"""
    import numpy as np
    rng = np.random.default_rng(5)
    seqs = ["".join(rng.choice(list("ACGTNacgt"), size=int(L), p=[.24, .24, .24, .24, .01, .0075, .0075, .0075, .0075]))
            for L in rng.integers(0, 700, size=5000)]
    fa = tmp_path / "r.fasta"
    fa.write_text("".join(">s%d\n%s\n" % (i, s) for i, s in enumerate(seqs)))
    bases, offsets = oracle.to_csr(seqs)
    for k in (3, 5):
        o = tmp_path / ("o%d" % k)
        assert run(cli, "comp", "oligo", "-i", fa, "-o", o, "-k", k).stderr == ""
        assert o.read_bytes() == oracle.oligo_text(oracle.oligo_batch(bases, offsets, k, True, True), True)
    d = tmp_path / "c"
    assert run(cli, "ctr", "-i", fa, "-o", d, "-k", "21").returncode == 0
    keys, counts = oracle.count_reads(bases, offsets, 21)
    assert sorted((d / "kmers.counts").read_text().splitlines()) == oracle.counts_lines(keys, counts)
    # cov -k 15 (table from the same file) and cov -a (table from another file), every preset
    cv = tmp_path / "cv"
    assert run(cli, "cov", "-i", fa, "-o", cv, "-k", "15", "-s", "5", "-c", "6").returncode == 0
    want_ctr = oracle.Counter(1)
    want_ctr.add_reads(bases, offsets, 15)
    assert (cv / "kmers.vectors").read_bytes() == oracle.oligo_text(want_ctr.cov_batch(bases, offsets, 15, 5, 6, True), True)
    k15, c15 = want_ctr.export()
    assert sorted((cv / "kmers.counts").read_text().splitlines()) == oracle.counts_lines(k15, c15)
    fa2 = tmp_path / "r2.fasta"
    fa2.write_text("".join(">s%d\n%s\n" % (i, s) for i, s in enumerate(seqs[:2000] * 3)))
    b2, o2 = oracle.to_csr(seqs[:2000] * 3)
    alt = oracle.Counter(1)
    alt.add_reads(b2, o2, 11)
    assert run(cli, "cov", "-i", fa, "-a", fa2, "-o", cv, "-k", "11", "--counts", "-p", "csv", "--bin-size=5").returncode == 0
    assert (cv / "kmers.vectors").read_bytes() == \
        oracle.oligo_text(alt.cov_batch(bases, offsets, 11, 5, 16, False), False).replace(b" ", b",")
    # whole-sequence CGR needs clean letters: strip N from the reads, vec-size 16
    clean = [s.replace("N", "") for s in seqs[:1500]]
    fa3 = tmp_path / "r3.fasta"
    fa3.write_text("".join(">s%d\n%s\n" % (i, s) for i, s in enumerate(clean)))
    w = tmp_path / "whole.cgr"
    assert run(cli, "comp", "cgr", "-i", fa3, "-o", w, "-v", "16").stderr == ""
    assert w.read_bytes() == oracle.cgr_text([oracle.cgr_points(s, 16) for s in clean])
    # min on the 5000 ragged reads: s2m lines in input order, m2s grouped by minimiser
    mo = tmp_path / "mins"
    recs = [("s%d" % i, s) for i, s in enumerate(seqs)]
    assert run(cli, "min", "-i", fa, "-o", mo, "-w", "20", "-m", "9").stderr == ""
    assert mo.read_text() == "".join(oracle.seq_to_min_lines(recs, 20, 9))
    long_enough = [(i, s) for i, s in recs if len(s) >= 10]
    fa4 = tmp_path / "r4.fasta"
    fa4.write_text("".join(">%s\n%s\n" % (i, s) for i, s in long_enough))
    assert run(cli, "min", "-i", fa4, "-o", mo, "-p", "m2s").stderr == ""
    assert sorted(mo.read_text().splitlines(keepends=True)) == sorted(oracle.bin_sequences_lines(long_enough, 0, 10))
    assert mo.read_text().splitlines() == sorted(mo.read_text().splitlines())


@pytest.mark.gpu
def test_ctr_out_of_core_passes_and_devices(cli, oracle, tmp_path):
    """a table 4x too small for the input's distinct k-mers: `ctr` runs hash-partition passes over the input
    (kt_ctr_add_reads_part; the reference spills chunks and merges partitions, counter/src/lib.rs:114-118,151-167,
    188-231) and kmers.counts is still the oracle's; `--devices 2` (two rank threads, here on one GPU through the
    in-process host all-to-all) gives the same lines; stdin is refused like the reference's SeqFormat::get("-")"""
    import os
    import numpy as np
    rng = np.random.default_rng(17)
    seqs = ["".join(rng.choice(list("ACGTN"), size=int(L), p=[.2475, .2475, .2475, .2475, .01]))
            for L in rng.integers(100, 600, size=6000)]
    seqs += ["A" * 3000, "ACGT" * 500] + seqs[:300]            # heavy hitters and repeats
    fq = tmp_path / "r.fastq"
    fq.write_text("".join("@s%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in enumerate(seqs)))
    bases, offsets = oracle.to_csr(seqs)
    keys, counts = oracle.count_reads(bases, offsets, 21, n_parts=4, threads=4)
    want = oracle.counts_lines(keys, counts)
    env = dict(os.environ, KT_CLI_TIMING="1", KT_BULK_MIN_BASES="0")
    d1 = tmp_path / "one"
    r = run(cli, "ctr", "-i", fq, "-o", d1, "-k", "21", env=env)
    assert r.returncode == 0 and ", 1 pass(es)" in r.stderr
    assert sorted((d1 / "kmers.counts").read_text().splitlines()) == want
    small = max(1024, int(1.4 * len(keys) / 4))                 # slots for a quarter of the distinct k-mers
    d4 = tmp_path / "four"
    r = run(cli, "ctr", "-i", fq, "-o", d4, "-k", "21", "-a", env=dict(env, KT_CTR_MAX_SLOTS=str(small)))
    assert r.returncode == 0, r.stderr
    passes = int(r.stderr.split(" pass(es)")[0].split()[-1])
    assert passes >= 4
    assert sorted((d4 / "kmers.counts").read_text().splitlines()) == oracle.counts_lines(keys, counts, k=21, acgt=True)
    # cov over a table that needs passes (coverage/src/lib.rs:69-184 works for any input its ctr can count): every pass's
    # table is probed for the k-mers of its hash partition, the raw rows summed, normalised at the end
    k15, c15 = oracle.count_reads(bases, offsets, 15, n_parts=4, threads=4)
    oc = oracle.Counter(1)
    oc.add_pairs(k15, c15)
    cv = tmp_path / "cov4"
    small15 = max(1024, int(1.4 * len(k15) / 4))
    for extra, norm, text in (((), True, None), (("--counts",), False, None)):
        r = run(cli, "cov", "-i", fq, "-o", cv, "-k", "15", "-s", "5", "-c", "6", *extra, env=dict(env, KT_CTR_MAX_SLOTS=str(small15)))
        assert r.returncode == 0, r.stderr
        assert int(r.stderr.split(" pass(es)")[0].split()[-1]) >= 4
        assert (cv / "kmers.vectors").read_bytes() == oracle.oligo_text(oc.cov_batch(bases, offsets, 15, 5, 6, norm), norm)
        assert sorted((cv / "kmers.counts").read_text().splitlines()) == oracle.counts_lines(k15, c15)
    d2 = tmp_path / "two"
    r = run(cli, "ctr", "-i", fq, "-o", d2, "-k", "21", "--devices", "2", env=dict(env, KT_CLI_SHARE_GPU="1"))
    assert r.returncode == 0, r.stderr
    assert sorted((d2 / "kmers.counts").read_text().splitlines()) == want
    r = run(cli, "ctr", "-i", "-", "-o", tmp_path / "x", "-k", "21", input=fq.read_text())
    assert r.returncode == 101 and "Error" in r.stderr
    # --devices must fail, not hang: a rank whose bring-up fails makes every rank stop before the first collective
    # (timeout of run() = the hang detector), and more devices than the node has are refused before any thread starts
    d3 = tmp_path / "three"
    r = run(cli, "ctr", "-i", fq, "-o", d3, "-k", "21", "--devices", "3", env=dict(env, KT_CLI_SHARE_GPU="1", KT_CLI_FAIL_RANK="1"))
    assert r.returncode != 0 and "bring-up" in r.stderr
    r = run(cli, "ctr", "-i", fq, "-o", d3, "-k", "21", "--devices", "63", env=env)
    assert r.returncode != 0 and "GPU(s)" in r.stderr
    # three ranks (the level-1 buckets do not divide evenly) give the oracle's lines too
    r = run(cli, "ctr", "-i", fq, "-o", d3, "-k", "21", "--devices", "3", env=dict(env, KT_CLI_SHARE_GPU="1"))
    assert r.returncode == 0, r.stderr
    assert sorted((d3 / "kmers.counts").read_text().splitlines()) == want
    # cov --devices N (coverage/src/lib.rs:69-184: the reference's cov builds its table with the same CountComputer, so
    # whatever ctr can count cov can look up): the table stays sharded on its GPUs, every batch of reads goes past every
    # shard (kt_cov_batch_part answers for the shard's own k-mers), the rows are summed - kmers.vectors and kmers.counts
    # are the oracle's, normalised and raw, with two and with three shards
    cvd = tmp_path / "covdev"
    for ndev, extra, norm in ((2, (), True), (3, ("--counts",), False)):
        r = run(cli, "cov", "-i", fq, "-o", cvd, "-k", "15", "-s", "5", "-c", "6", "--devices", str(ndev), *extra,
                env=dict(env, KT_CLI_SHARE_GPU="1"))
        assert r.returncode == 0, r.stderr
        assert (cvd / "kmers.vectors").read_bytes() == oracle.oligo_text(oc.cov_batch(bases, offsets, 15, 5, 6, norm), norm)
        assert sorted((cvd / "kmers.counts").read_text().splitlines()) == oracle.counts_lines(k15, c15)


@pytest.mark.gpu
def test_ctr_out_of_core_bounded_host_memory(cli, oracle, tmp_path):
    """the reference's whole design for `ctr` is a memory ceiling (chunks sized by -m, counter/src/lib.rs:114-118; the
    merged map streamed straight into the file, :220-230).  Here a pass's table is staged on the device and fetched a slab
    of a million entries at a time (kt_ctr_export_stage / _fetch) by ONE writer, from ONE reader that is rewound per pass,
    so what the host holds does not depend on how many distinct k-mers a pass has or on how many passes there are: ~84 M
    distinct 31-mers counted in >= 4 passes (>= 21 M entries = 250 MB of (key, count) pairs per pass, were they held
    whole) and in >= 16 passes, the lines are the oracle's.

    What is asserted comes from the process's OWN books, printed per pass under KT_CLI_TIMING (computers.cpp): the
    capacities of its buffers (reader window, batch, slab of pairs, text pieces) and the allocator's total
    (mallinfo2: arenas in use + its mmaps) at every slab boundary.  Round 4 compared the VmHWM of the two processes to 64 MB
    - but three quarters of a gigabyte of either peak are the HIP runtime's mappings, which vary by more than that from
    box to box (787 MB and ~1 GB for the same run); VmHWM is still printed and bounded, with room for that variance."""
    import os
    import re
    import numpy as np
    n, L, k = 700_000, 150, 31
    hb, ho = oracle.synth_reads(31337, n, L, noise=False)
    fa = tmp_path / "big.fa"
    with open(fa, "wb") as f:
        for i in range(n):
            f.write(b">r%d\n" % i)
            f.write(hb[int(ho[i]):int(ho[i + 1])].tobytes())
            f.write(b"\n")
    keys, counts = oracle.count_reads(hb, ho, k, n_parts=8, threads=8)
    assert len(keys) > 80_000_000
    env = dict(os.environ, KT_CLI_TIMING="1", KT_CLI_BATCH_BASES=str(8 << 20))
    per_pass = re.compile(r"pass (\d+) written: entries (\d+), own buffers (\d+) kB, malloc in use (\d+) kB \(peak at a slab boundary "
                          r"(\d+) kB\), malloc free \d+ kB, VmRSS (\d+) kB .*VmHWM (\d+) kB")
    books = {}
    for passes in (4, 16):
        small = max(1024, int(1.4 * len(keys) / passes))
        out = tmp_path / ("p%d" % passes)
        r = run(cli, "ctr", "-i", fa, "-o", out, "-k", str(k), "-m", "6", env=dict(env, KT_CTR_MAX_SLOTS=str(small)))
        assert r.returncode == 0, r.stderr
        n_passes = int(r.stderr.split(" pass(es)")[0].split()[-1])
        assert n_passes >= passes
        rows = np.array([[int(x) for x in m.groups()] for m in per_pass.finditer(r.stderr)], dtype=np.int64)
        assert len(rows) == n_passes and int(rows[:, 1].sum()) == len(keys), r.stderr[-3000:]
        books[passes] = rows
        print("passes %d: entries per pass <= %d, own buffers %d..%d kB, malloc in use %d..%d kB, slab-boundary peak %d kB, "
              "VmRSS %d..%d kB, VmHWM %d kB" % (n_passes, rows[:, 1].max(), rows[:, 2].min(), rows[:, 2].max(), rows[:, 3].min(),
                                                rows[:, 3].max(), rows[:, 4].max(), rows[:, 5].min(), rows[:, 5].max(), rows[:, 6].max()))
        if passes == 4:
            got = np.loadtxt(out / "kmers.counts", dtype=np.uint64, delimiter="\t")
            order = np.argsort(got[:, 0])
            assert np.array_equal(got[order, 0], keys) and np.array_equal(got[order, 1], counts.astype(np.uint64))
            del got, order
        (out / "kmers.counts").unlink()
    MB = 1024
    a, b = books[4], books[16]
    # a pass of the first run holds > 3x the entries of a pass of the second ...
    assert a[:, 1].max() > 3 * b[:, 1].max() and a[:, 1].max() * 12 > 120 * MB * 1024
    for rows in (a, b):
        # ... and neither holds more than a fixed set of buffers: the program's own <= 224 MB (measured 145-171: the reader's
        # window of parsed pieces ~100, a batch 8, the writer's slab of pairs 12 + its text 32) where the table is 1 GB of
        # pairs + 1.9 GB of text; the allocator's total at a slab boundary <= 288 MB
        assert rows[:, 2].max() <= 224 * MB and rows[:, 4].max() <= 288 * MB, rows
        # nothing grows with the pass index: from the end of the second pass (the reader's buffer pool has filled by then)
        # to the end of the last one, and at any pass in between
        assert rows[-1, 3] - rows[1, 3] <= 32 * MB and rows[:, 3].max() - rows[1, 3] <= 32 * MB, rows
        assert rows[-1, 5] - rows[1, 5] <= 64 * MB, rows
        # the kernel's view (HIP runtime included; 0.78-1.0 GB on the boxes seen)
        assert rows[:, 6].max() <= 1536 * MB, rows
    # the same books whatever the number of passes
    assert abs(int(a[:, 2].max()) - int(b[:, 2].max())) <= 48 * MB and abs(int(a[:, 4].max()) - int(b[:, 4].max())) <= 48 * MB


def test_parallel_reader_matches_serial(cli, tmp_path):
    """plain files of >= 32 MB are mapped and parsed piecewise by several threads: the records must be exactly the
    serial reader's (KT_READER_THREADS=1), in order, with their ordinals - multi-line FASTA with CRLF and blank
    lines, and FASTQ whose quality lines begin with '@' or '+' (the classic trap for boundary detection)"""
    import hashlib
    import os
    import numpy as np
    rng = np.random.default_rng(3)
    alpha = np.frombuffer(b"ACGTNacgt", np.uint8)
    qual = np.frombuffer(b"@+IIIIFFFF#5", np.uint8)

    def digest(path, threads):
        env = dict(os.environ, KT_READER_THREADS=str(threads))
        r = subprocess.run([cli, "debug-read", str(path)], capture_output=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-500:]
        lines = r.stdout.splitlines()
        meta = [ln for ln in lines if ln.startswith(b"#")]
        return hashlib.sha256(r.stdout).hexdigest(), len(lines) - len(meta), meta

    fa = tmp_path / "big.fasta"
    with open(fa, "wb") as f:
        for i in range(90_000):
            L = int(rng.integers(0, 900))
            s = alpha[rng.integers(0, 9, size=L)].tobytes()
            eol = b"\r\n" if i % 7 == 0 else b"\n"
            f.write(b">r%d some text%s" % (i, eol))
            for j in range(0, L, 70):
                f.write(s[j:j + 70] + eol)
            if i % 11 == 0:
                f.write(eol)
    fq = tmp_path / "big.fastq"
    with open(fq, "wb") as f:
        for i in range(140_000):
            L = int(rng.integers(1, 260))
            s = alpha[rng.integers(0, 4, size=L)].tobytes()
            q = qual[rng.integers(0, len(qual), size=L)].tobytes()
            f.write(b"@q%d extra\n%s\n+\n%s\n" % (i, s, q))
    # a FASTQ whose sequences and qualities are wrapped over several lines (quality lines starting with '@'): the serial
    # parser reads it; the piecewise reader's 4-line boundary rule does not hold, so the file must stay with the former
    fw = tmp_path / "wrapped.fastq"
    with open(fw, "wb") as f:
        for i in range(60_000):
            L = int(rng.integers(100, 900))
            s = alpha[rng.integers(0, 4, size=L)].tobytes()
            q = b"@" + b"I" * (L - 1)
            f.write(b"@w%d\n" % i)
            for j in range(0, L, 80):
                f.write(s[j:j + 80] + b"\n")
            f.write(b"+\n")
            for j in range(0, L, 80):
                f.write(q[j:j + 80] + b"\n")
    assert fa.stat().st_size > (32 << 20) and fq.stat().st_size > (32 << 20) and fw.stat().st_size > (32 << 20)
    for path, n in ((fa, 90_000), (fq, 140_000), (fw, 60_000)):
        serial = digest(path, 1)
        assert serial[1] == n
        for threads in (2, 5):
            assert digest(path, threads) == serial
    # rewind() (the passes of an out-of-core count read ONE reader again and again: its threads, cuts and buffers stay):
    # three passes - the second left after 1000 records, with pieces parsed ahead and never taken - are the serial
    # reader's records each time, ordinals from 0
    for path, n in ((fa, 90_000), (fq, 140_000), (fw, 60_000)):
        one = subprocess.run([cli, "debug-read", str(path)], capture_output=True, env=dict(os.environ, KT_READER_THREADS="1"),
                             timeout=600).stdout
        body = one[:one.index(b"#records")]
        for threads in (1, 4):
            r = subprocess.run([cli, "debug-read", str(path)], capture_output=True, timeout=600,
                               env=dict(os.environ, KT_READER_THREADS=str(threads), KT_DEBUG_READ_PASSES="3"))
            assert r.returncode == 0, r.stderr[-500:]
            p0, rest = r.stdout.split(b"#pass\t1\n")
            p1, p2 = rest.split(b"#pass\t2\n")
            assert p0 == body and p2[:p2.index(b"#records")] == body and p2[p2.index(b"#records"):] == one[one.index(b"#records"):]
            assert 1000 <= p1.count(b"\n") < 1010 and body.startswith(p1)


# ---- several batches, several threads, repeated: the reference's own end-to-end idiom ----------------------------
def _write_reads(path, n, seed, fastq=False, genome=0, chunk=100_000):
    """n ragged reads (40..260 bases; N, lower case; genome > 0: sampled from a random genome of that length so that
    k-mers repeat) as FASTA or FASTQ with ids r0000000...; -> (bases u8[total], offsets u64[n + 1]).  Some FASTQ
    quality lines begin with '@' (legal, and the record-boundary finder of the parallel reader has to cope)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    qrng = np.random.default_rng(seed + 1000)   # (its own stream: FASTA and FASTQ of one seed hold the same reads)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    gen = alpha[rng.integers(0, 4, size=genome)] if genome else None
    all_bases, all_lens = [], []
    with open(path, "wb") as fh:
        for c0 in range(0, n, chunk):
            m = min(chunk, n - c0)
            lens = rng.integers(40, 261, size=m)
            lens[rng.random(m) < 0.001] = 0                     # a few empty records
            off = np.zeros(m + 1, np.int64)
            np.cumsum(lens, out=off[1:])
            T = int(off[-1])
            if genome:
                start = rng.integers(0, genome - 260, size=m)
                idx = np.repeat(start - off[:-1], lens) + np.arange(T)
                b = gen[idx].copy()
                err = rng.random(T) < 0.01
                b[err] = alpha[rng.integers(0, 4, size=int(err.sum()))]
            else:
                b = alpha[rng.integers(0, 4, size=T)]
            r = rng.random(T)
            b[r < 0.002] = ord("N")
            b[(r > 0.002) & (r < 0.02)] |= 0x20
            # records: ">r0000000\n" + seq + "\n"   or   "@r0000000\n" + seq + "\n+\n" + qual + "\n"
            ids = np.arange(c0, c0 + m)
            hdr = np.empty((m, 10), np.uint8)
            hdr[:, 0] = ord("@" if fastq else ">")
            hdr[:, 1] = ord("r")
            for d in range(7):
                hdr[:, 8 - d] = ord("0") + (ids // 10 ** d) % 10
            hdr[:, 9] = ord("\n")
            per = 10 + lens + 1 + ((2 + lens + 1) if fastq else 0)
            rs = np.zeros(m + 1, np.int64)
            np.cumsum(per, out=rs[1:])
            out = np.full(int(rs[-1]), ord("\n"), np.uint8)
            out[(rs[:-1, None] + np.arange(10)[None, :]).ravel()] = hdr.ravel()
            rid = np.repeat(np.arange(m), lens)
            within = np.arange(T) - off[:-1][rid]
            out[rs[:-1][rid] + 10 + within] = b
            if fastq:
                out[rs[:-1] + 10 + lens + 1] = ord("+")
                q = np.full(T, ord("I"), np.uint8)
                first = off[:-1][lens > 0]
                q[first[qrng.random(first.size) < 0.02]] = ord("@")
                out[rs[:-1][rid] + 10 + lens[rid] + 3 + within] = q
            fh.write(out.tobytes())
            all_bases.append(b)
            all_lens.append(lens)
    lens = np.concatenate(all_lens)
    offsets = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=offsets[1:])
    return np.concatenate(all_bases), offsets


def _same_file(a, b):
    import filecmp
    return filecmp.cmp(str(a), str(b), shallow=False)


@pytest.mark.gpu
def test_cli_many_batches_many_threads_repeated(cli, oracle, tmp_path):
    """composition/src/oligo.rs:327-368: the reference's tests push a file through several batches with 8 threads and
    repeat the run, comparing the bytes.  Here: 1.4 M ragged reads (4 batches of 64 Mbases through the rotating work
    items, the pinned row buffers, the ordered writer), FASTA and FASTQ, -t 1 and -t 8, the parallel reader on and off,
    `comp oligo -k 4` normalised (three times over) and -c against the oracle's text byte for byte; then `comp cgr
    -k 5`, `ctr -k 21` and `cov -k 15` over many (forced) batches of reads that share k-mers."""
    import os
    import numpy as np
    n = 1_400_000
    fa, fq = tmp_path / "big.fa", tmp_path / "big.fq"
    bases, offsets = _write_reads(fa, n, 21)
    assert int(offsets[-1]) > 3 * (64 << 20)                      # more than three 64-Mbase batches
    cores = min(os.cpu_count() or 8, 32)
    want_n, want_c = tmp_path / "want_norm", tmp_path / "want_counts"
    for r0 in range(0, n, 120_000):                                # the oracle's rows, a slab at a time
        r1 = min(n, r0 + 120_000)
        b, o = bases[int(offsets[r0]):int(offsets[r1])], offsets[r0:r1 + 1] - offsets[r0]
        oracle.matrix_text_file(oracle.oligo_batch(b, o, 4, True, True, threads=cores), want_n, "fixed6",
                                append=r0 > 0, threads=cores)
        oracle.matrix_text_file(oracle.oligo_batch(b, o, 4, True, False, threads=cores), want_c, "display",
                                append=r0 > 0, threads=cores)
    out = tmp_path / "out"
    serial = dict(os.environ, KT_READER_THREADS="1")

    def oligo(src, *extra, env=None):
        r = run(cli, "comp", "oligo", "-i", src, "-o", out, "-k", 4, *extra, env=env)
        assert r.returncode == 0 and r.stderr == "", r.stderr
    for rep in range(3):                                           # the reference's repetition idiom
        oligo(fa, "-t", 8)
        assert _same_file(out, want_n), "normalised, -t 8, parallel reader, repeat %d" % rep
    oligo(fa, "-t", 1, env=serial)
    assert _same_file(out, want_n), "normalised, -t 1, serial reader"
    oligo(fa, "-t", 8, "-c", env=serial)
    assert _same_file(out, want_c), "-c, -t 8, serial reader"
    _write_reads(fq, n, 21, fastq=True)                            # the same reads as FASTQ
    oligo(fq, "-t", 1, "-c")
    assert _same_file(out, want_c), "-c, -t 1, FASTQ, parallel reader"
    oligo(fq, "-t", 8)
    assert _same_file(out, want_n), "normalised, -t 8, FASTQ, parallel reader"
    for p in (out, want_n, want_c, fq, fa):
        p.unlink()

    # reads that share k-mers (37x over a 3 Mbp genome), pushed through many small batches
    m = 420_000
    g = tmp_path / "genome_reads.fa"
    gb, go = _write_reads(g, m, 22, genome=3_000_000)
    assert g.stat().st_size > (32 << 20)                           # big enough for the parallel reader
    import pandas as pd
    small = dict(os.environ, KT_CLI_BATCH_BASES=str(13 << 20), KT_CLI_BATCH_READS="70000")
    keys, counts = oracle.count_reads(gb, go, 21, n_parts=cores, threads=cores)
    oc = oracle.Counter(cores)
    oc.add_reads(gb, go, 15, threads=cores)
    want_v = tmp_path / "want_cov"
    oracle.matrix_text_file(oc.cov_batch(gb, go, 15, 16, 16, True), want_v, "fixed6", threads=cores)
    del oc
    for env in (small, dict(small, KT_READER_THREADS="1")):
        d = tmp_path / "ctr"
        r = run(cli, "ctr", "-i", g, "-o", d, "-k", 21, "-t", 8, env=env)
        assert r.returncode == 0, r.stderr
        got = pd.read_csv(d / "kmers.counts", sep="\t", header=None, dtype=np.uint64).to_numpy()
        order = np.argsort(got[:, 0], kind="stable")
        assert np.array_equal(got[order, 0], keys) and np.array_equal(got[order, 1], counts.astype(np.uint64))
        del got
        cv = tmp_path / "cov"
        r = run(cli, "cov", "-i", g, "-o", cv, "-k", 15, "-t", 8, env=env)
        assert r.returncode == 0, r.stderr
        assert _same_file(cv / "kmers.vectors", want_v)
    sub = 60_000
    s = tmp_path / "sub.fa"
    sb, so = _write_reads(s, sub, 23, genome=3_000_000)
    want_g = tmp_path / "want_cgr"
    oracle.matrix_text_file(oracle.oligo_batch(sb, so, 5, True, True, threads=cores), want_g, "cgr",
                            xy=oracle.cgr_coords(5, 25), threads=cores)   # (vec-size defaults to k^2, args.rs:266-269)
    for t in (1, 8):
        r = run(cli, "comp", "cgr", "-i", s, "-o", out, "-k", 5, "-t", t, env=dict(os.environ, KT_CLI_BATCH_READS="9000"))
        assert r.returncode == 0 and r.stderr == "", r.stderr
        assert _same_file(out, want_g), "comp cgr -k 5, -t %d, 7 batches" % t
