"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/kmertools_hip.h
declares, its host-side helpers agree with the oracle, and compute entry points fail loudly
(no CPU fallback) when no GPU is present."""
import ctypes
import pathlib
import re

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def klib():
    from kmertools_amd import _lib
    if not _lib.LIB_PATH.exists():
        _lib.build()
    return _lib


def test_header_symbols_exported(klib):
    header = (ROOT / "include" / "kmertools_hip.h").read_text()
    declared = set(re.findall(r"\b(kt_[a-z0-9_]+)\s*\(", header))
    declared -= {"kt_ctx", "kt_ctr"}
    assert declared, "no declarations parsed"
    raw = ctypes.CDLL(str(klib.LIB_PATH))
    for name in sorted(declared):
        assert hasattr(raw, name), "libkmertools_hip.so does not export %s" % name
    # and the python binding covers all of them
    assert declared == set(klib.SYMBOLS), declared ^ set(klib.SYMBOLS)
    assert klib.lib().kt_version() >= 100
    # ... and INTEGRATION.md shows the reference-side (Rust) binding line of every one of them
    integ = (ROOT / "INTEGRATION.md").read_text()
    unbound = [name for name in sorted(declared) if "pub fn %s(" % name not in integ]
    assert not unbound, "INTEGRATION.md has no extern \"C\" line for %s" % unbound


def test_host_helpers_match_oracle(klib, oracle):
    from kmertools_amd import device
    for k in range(1, 8):
        m, pk, c = device.pos_map(k)
        wm, wpk, wc = oracle.pos_maps(k)
        assert c == wc and np.array_equal(m.astype(np.uint64), wm) and np.array_equal(pk, wpk)
        assert device.bins(k, True) == wc and device.bins(k, False) == 4 ** k
    rng = np.random.default_rng(1)
    for k in (1, 2, 5, 15, 16, 31, 32):
        for x in rng.integers(0, 2 ** 62, size=200, dtype=np.uint64):
            v = int(x) & ((1 << (2 * k)) - 1)
            assert device.rev_comp(v, k) == oracle.rev_comp(v, k)
            s = oracle.numeric_to_kmer(v, k)
            assert device.numeric_to_kmer(v, k) == s
            if k <= 31:
                assert device.kmer_to_numeric(s) == oracle.kmer_to_numeric(s) == (v, oracle.rev_comp(v, k))
    # no validity check, like the reference (kmer/src/lib.rs:42-47)
    assert device.kmer_to_numeric("ACNGT") == oracle.kmer_to_numeric("ACNGT")
    for k, vs in [(3, 9), (4, 16), (5, 25), (7, 49), (4, 1)]:
        assert np.array_equal(device.cgr_coords(k, vs), oracle.cgr_coords(k, vs))


def test_kat_host(klib, kat):
    from kmertools_amd import device
    for c in kat["rev_comp"]:
        assert device.rev_comp(c["kmer"], c["k"]) == c["rc"]
    for v, k, s in kat["numeric"]["to_acgt"]:
        assert device.numeric_to_kmer(v, k) == s
    for s, f, r in kat["numeric"]["to_numeric"]:
        assert device.kmer_to_numeric(s) == (f, r)
    m, pk, cnt = device.pos_map(4)
    e = kat["pos_map_k4"]
    assert cnt == e["count"] and int((m > 0).sum()) == e["nonzero_entries"]
    for idx, v in e["entries"].items():
        assert int(m[int(idx)]) == v


def test_error_reporting(klib):
    from kmertools_amd import device
    with pytest.raises(klib.KmertoolsError) as ei:
        device.bins(99)
    assert ei.value.code == klib.KT_ERR_ARG and "k must be" in str(ei.value)
    with pytest.raises(klib.KmertoolsError):
        device.kmer_to_numeric("A" * 33)


def test_owner_of_is_a_partition(klib):
    from kmertools_amd import device
    rng = np.random.default_rng(3)
    keys = rng.integers(0, 2 ** 62, size=4000, dtype=np.uint64)
    for n in (1, 2, 3, 8, 64):
        owners = np.array([device.owner_of(int(x), n) for x in keys])
        assert owners.min() >= 0 and owners.max() < n
        if n > 1:
            cnt = np.bincount(owners, minlength=n)
            assert cnt.min() > 0.5 * len(keys) / n     # roughly balanced


def test_no_cpu_fallback(klib):
    """Without a GPU the product path must raise, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from kmertools_amd import device, pykmertools
    with pytest.raises(klib.KmertoolsError) as ei:
        device.Context(0)
    assert ei.value.code == klib.KT_ERR_NODEVICE
    with pytest.raises(klib.KmertoolsError):
        pykmertools.OligoComputer(4).vectorise_one("ACGTACGT")


def test_product_does_not_import_oracle():
    for p in (ROOT / "kmertools_amd").rglob("*"):
        if p.suffix in (".py", ".hip", ".cpp", ".hpp", ".h"):
            txt = p.read_text()
            assert "kt_oracle" not in txt and "libkt_oracle" not in txt, p
            assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), p
