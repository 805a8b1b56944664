"""`import pykmertools as kt`: the reference's Python module by name (pip/src/lib.rs:31-40).

The bodies below restate the reference's own tests/test_{oligo,cgr,kmers,min,utils}.py (Biopython is not in this
image, so FASTQ is read by the oracle's small parser; `eval` of the fixture's tuples is kept as the reference has it),
then go past them: ksize outside the CLI's 3..7 (pybindings/src/oligo.rs:22-31 takes any), run_cli (pip/src/lib.rs:11-18).
"""
import pathlib
import subprocess
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]


def test_module_surface_by_name():
    """no GPU needed: the package imports and offers exactly what the reference registers"""
    import pykmertools as kt
    from pykmertools import utils as ktutils
    for name in ("OligoComputer", "CgrComputer", "KmerGenerator", "MinimiserGenerator", "run_cli", "utils"):
        assert hasattr(kt, name), name
    assert callable(ktutils.to_acgt) and callable(ktutils.to_numeric)
    assert ktutils.to_acgt(111, 5) == "ACGTT" and ktutils.to_acgt(27, 5) == "AACGT"     # tests/test_utils.py:4-9
    assert ktutils.to_numeric("ACGTT") == (111, 27)                                      # tests/test_utils.py:12-16
    with pytest.raises(ValueError):
        ktutils.to_numeric("A" * 33)                                                     # pybindings/src/kmer.rs:58-63


@pytest.mark.gpu
def test_reference_test_oligo(oracle, golden):
    import pykmertools as kt
    oligo_gen = kt.OligoComputer(4)
    seqs = [s.decode() for _, s in oracle.read_records(golden / "reads.fq")]
    oligos_generated = [list(map(lambda x: round(x, 6), line)) for line in oligo_gen.vectorise_batch(seqs)]
    oligos_truth = [list(map(float, line.strip().split())) for line in (golden / "expected_fa.kmers").read_text().splitlines()]
    assert len(oligos_generated) == len(oligos_truth) == 2
    for g, t in zip(oligos_generated, oligos_truth):
        assert g == t
    header_generated = oligo_gen.get_header()
    assert len(header_generated) == 136
    assert len(oligo_gen.get_header(False)) == 256


@pytest.mark.gpu
def test_reference_test_cgr(oracle, golden):
    import pykmertools as kt
    cgr_gen = kt.CgrComputer(1)
    seqs = [s.decode() for _, s in oracle.read_records(golden / "reads.fq")]
    cgrs_generated = cgr_gen.vectorise_batch(seqs)
    cgrs_truth = [[eval(item) for item in line.split(" ")] for line in (golden / "expected_reads.cgr").read_text().splitlines()]
    assert len(cgrs_generated) == len(cgrs_truth)
    for g, t in zip(cgrs_generated, cgrs_truth):
        assert g == t


@pytest.mark.gpu
def test_reference_test_kmers_and_min():
    import pykmertools as kt
    from pykmertools import utils as ktutils
    kmers = list(kt.KmerGenerator("ACGTCC", 3))
    for (fmer, _), acgt_mer in zip(kmers, ["ACG", "CGT", "GTC", "TCC"]):
        assert ktutils.to_acgt(fmer, len(acgt_mer)) == acgt_mer
    assert len(kmers) == 4
    min_gen = kt.MinimiserGenerator(
        "ATGCGATATCGTAGGCGTCGATGGAGAGCTAGATCGATCGATCTAAATCCCGATCGATTCCGAGCGCGATCAAAGCGCGATAGGCTAGCTAAAGCTAGCA", 31, 7)
    mins = ["ACGATAT", "ACGCCTA", "AGAGCTA", "AAATCCC", "AATCCCG", "AATCGAT", "AAAGCGC"]
    got = list(min_gen)
    assert len(got) == len(mins)
    for (kmer, _, _), m in zip(got, mins):
        assert min_gen.to_acgt(kmer) == m


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 8, 9, 10, 12])
def test_oligo_any_ksize(oracle, k):
    """ksize outside 3..7 takes the global-memory histogram path: bit-identical f64 rows, both modes, the raw-mode
    total += 2 quirk of pybindings/src/oligo.rs:61 included"""
    import pykmertools as kt
    rng = np.random.default_rng(k)
    alpha = np.frombuffer(b"ACGTNacgt", np.uint8)
    n = 40 if k <= 10 else 3
    seqs = [alpha[rng.integers(0, 9 if i % 3 == 0 else 4, size=int(rng.integers(0, 400)))].tobytes().decode() for i in range(n)]
    seqs[1] = ""
    seqs[2] = "ACGT"[: max(0, k - 1)]
    gen = kt.OligoComputer(k)
    bases, offsets = oracle.to_csr([s.encode() for s in seqs])
    for mins in (True, False):
        if not mins and k > 10:
            continue   # 4^12 raw bins x 8 bytes per read: skipped for time, not for capability
        for norm in (True, False):
            got = gen.vectorise_batch_numpy(seqs, norm=norm, mins=mins)
            want = oracle.oligo_batch(bases, offsets, k, mins, norm, 1.0 if mins else 2.0)
            assert got.shape == want.shape
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert len(gen.get_header()) == got.shape[1] if mins else True


@pytest.mark.gpu
def test_run_cli_is_the_command_line(golden, tmp_path):
    """run_cli = the kmertools CLI with the process's arguments (pip/src/lib.rs:11-18), here through `python -m
    pykmertools`: byte-identical to the reference's fixture"""
    out = tmp_path / "fa.kmers"
    r = subprocess.run([sys.executable, "-m", "pykmertools", "comp", "oligo", "-i", str(golden / "reads.fa"), "-o", str(out),
                        "-k", "4"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == (golden / "expected_fa.kmers").read_bytes()
