"""End-to-end CLI wall clock (file -> reader -> H2D -> kernels -> D2H -> text -> file) on the GPU box.
usage: python tools/cli_e2e.py [n_reads] [read_len]   (writes under /tmp; prints one line per command)"""
import os, subprocess, sys, time, pathlib
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
root = pathlib.Path(__file__).resolve().parents[1]
cli = root / "kmertools_amd" / "bin" / "kmertools"
tmp = pathlib.Path(os.environ.get("TMPDIR", "/tmp")) / "kt_e2e"
tmp.mkdir(exist_ok=True)
fa = tmp / "reads.fa"
rng = np.random.default_rng(1)
genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=20_000_000)]
starts = rng.integers(0, len(genome) - L, size=n)
rec = np.empty((n, L + 12), np.uint8)           # ">" + 9-digit id + "\n" + L bases + "\n"
rec[:, 0] = ord(">")
ids = np.char.zfill(np.arange(n).astype(str), 9)
rec[:, 1:10] = np.frombuffer("".join(ids).encode(), np.uint8).reshape(n, 9)
rec[:, 10] = ord("\n")
rec[:, 11:11 + L] = genome[starts[:, None] + np.arange(L)[None, :]]
rec[:, 11 + L] = ord("\n")
fa.write_bytes(rec.tobytes())
print("input: %d reads x %d bp, %.1f MB FASTA" % (n, L, fa.stat().st_size / 1e6), flush=True)
env = dict(os.environ, KT_CLI_TIMING="1")
cmds = [
    ("comp oligo k=4", ["comp", "oligo", "-i", fa, "-o", tmp / "o4.kmers", "-k", "4"]),
    ("comp oligo k=4 counts", ["comp", "oligo", "-i", fa, "-o", tmp / "o4c.kmers", "-k", "4", "-c"]),
    ("ctr k=31", ["ctr", "-i", fa, "-o", tmp / "c31", "-k", "31"]),
    ("cov k=15", ["cov", "-i", fa, "-o", tmp / "v15", "-k", "15"]),
    ("comp cgr (whole seq)", ["comp", "cgr", "-i", fa, "-o", tmp / "whole.cgr"]),
    ("min w=31 m=7 s2m", ["min", "-i", fa, "-o", tmp / "mins.s2m", "-w", "31", "-m", "7"]),
    ("min w=0 m=10 m2s", ["min", "-i", fa, "-o", tmp / "mins.m2s", "-p", "m2s"]),
]
only = os.environ.get("ONLY")
for name, args in cmds:
    if only and only not in name:
        continue
    t0 = time.perf_counter()
    r = subprocess.run([str(cli)] + [str(a) for a in args], env=env, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    print("%-24s %.2f s  %.3f Gbases/s  rc=%d" % (name, dt, n * L / dt / 1e9, r.returncode), flush=True)
    for ln in r.stderr.splitlines():
        print("    " + ln)
for f in tmp.rglob("*"):
    if f.is_file():
        print("    %s %.1f MB" % (f.relative_to(tmp), f.stat().st_size / 1e6))
        f.unlink()
