"""Sanitizers on the CPU builds (GPU sanitizers are not available on this pool): the oracle's threaded
counter / spill / merge / histogram / text workers under AddressSanitizer + UBSan and ThreadSanitizer
(oracle/sanitize_driver.c), and the C++ host side - the parallel reader on >= 32 MB multi-thread fixtures
(`debug-read`), the {:.6} formatter (`debug-fixed6`) and the formatter workers + ordered writer (`debug-emit`) -
in its asan / tsan builds.  Everything here runs without a GPU."""
import hashlib
import os
import pathlib
import subprocess

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
CSRC = ROOT / "kmertools_amd" / "csrc"
BIN = ROOT / "kmertools_amd" / "bin"
SAN_ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0")


def _clean(stderr):
    text = stderr.decode("utf-8", "replace") if isinstance(stderr, bytes) else stderr
    assert "AddressSanitizer" not in text and "ThreadSanitizer" not in text and "runtime error" not in text, text[-3000:]


@pytest.fixture(scope="module")
def san_builds():
    subprocess.check_call(["make", "-C", str(ROOT / "oracle"), "asan", "tsan"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", str(CSRC), "asan", "tsan"], stdout=subprocess.DEVNULL)
    return {"oracle_asan": ROOT / "oracle" / "kt_oracle_asan", "oracle_tsan": ROOT / "oracle" / "kt_oracle_tsan",
            "cli": BIN / "kmertools", "cli_asan": BIN / "kmertools_asan", "cli_tsan": BIN / "kmertools_tsan"}


@pytest.mark.parametrize("which,reads", [("oracle_asan", 12000), ("oracle_tsan", 4000)])
def test_oracle_threads_under_sanitizers(san_builds, tmp_path, which, reads):
    r = subprocess.run([str(san_builds[which]), str(tmp_path), str(reads), "8"], capture_output=True, env=SAN_ENV,
                       timeout=900)
    _clean(r.stderr)
    assert r.returncode == 0, r.stderr[-2000:]
    assert b"sanitize_driver ok" in r.stdout


@pytest.fixture(scope="module")
def big_fixtures(tmp_path_factory):
    """the >= 32 MB fixtures the parallel reader needs: multi-line FASTA with CRLF / blank lines, FASTQ whose quality
    lines begin with '@' or '+'"""
    d = tmp_path_factory.mktemp("san")
    rng = np.random.default_rng(4)
    alpha = np.frombuffer(b"ACGTNacgt", np.uint8)
    qual = np.frombuffer(b"@+IIIIFFFF#5", np.uint8)
    fa, fq = d / "big.fasta", d / "big.fastq"
    with open(fa, "wb") as f:
        for i in range(80_000):
            L = int(rng.integers(0, 900))
            s = alpha[rng.integers(0, 9, size=L)].tobytes()
            eol = b"\r\n" if i % 7 == 0 else b"\n"
            f.write(b">r%d some text%s" % (i, eol))
            for j in range(0, L, 70):
                f.write(s[j:j + 70] + eol)
            if i % 11 == 0:
                f.write(eol)
    with open(fq, "wb") as f:
        for i in range(140_000):
            L = int(rng.integers(1, 260))
            s = alpha[rng.integers(0, 4, size=L)].tobytes()
            q = qual[rng.integers(0, len(qual), size=L)].tobytes()
            f.write(b"@q%d extra\n%s\n+\n%s\n" % (i, s, q))
    assert fa.stat().st_size > (32 << 20) and fq.stat().st_size > (32 << 20)
    return fa, fq


@pytest.mark.parametrize("which", ["cli_asan", "cli_tsan"])
def test_host_reader_and_formatters_under_sanitizers(san_builds, big_fixtures, which):
    exe = str(san_builds[which])

    def digest(binary, path, threads):
        r = subprocess.run([binary, "debug-read", str(path)], capture_output=True, timeout=1500,
                           env=dict(SAN_ENV, KT_READER_THREADS=str(threads)))
        _clean(r.stderr)
        assert r.returncode == 0, r.stderr[-2000:]
        return hashlib.sha256(r.stdout).hexdigest()

    for path in big_fixtures:
        want = digest(str(san_builds["cli"]), path, 1)            # the plain build's serial reader
        assert digest(exe, path, 4) == want                      # four parser threads under the sanitizer
    r = subprocess.run([exe, "debug-fixed6", "200000"], capture_output=True, env=SAN_ENV, timeout=900)
    _clean(r.stderr)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-500:]
    r = subprocess.run([exe, "debug-emit", "20000", "136", "1", "8", "2"], capture_output=True, env=SAN_ENV, timeout=900)
    _clean(r.stderr)
    assert r.returncode == 0, r.stderr[-2000:]
