import os, subprocess, sys, pathlib, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import test_cli as t
from oracle import kt_oracle as oracle
d = pathlib.Path(tempfile.mkdtemp())
s = d / "sub.fa"
sb, so = t._write_reads(s, 60000, 23, genome=3_000_000)
want = d / "want"
oracle.matrix_text_file(oracle.oligo_batch(sb, so, 5, True, True, threads=16), want, "cgr", xy=oracle.cgr_coords(5, 1), threads=16)
for env in ({}, {"KT_CLI_BATCH_READS": "9000"}):
    out = d / "out"
    r = subprocess.run([str(t.CLI), "comp", "cgr", "-i", str(s), "-o", str(out), "-k", "5", "-t", "1"], capture_output=True, text=True, env=dict(os.environ, **env))
    print(env, r.returncode, r.stderr[:200])
    a = open(out, "rb").read().split(b"\n"); b = open(want, "rb").read().split(b"\n")
    print("lines", len(a), len(b))
    nd = 0
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            xs, ys = x.split(b" "), y.split(b" ")
            for j, (p, q) in enumerate(zip(xs, ys)):
                if p != q:
                    print("line", i, "field", j, "cli", p, "oracle", q, "len", int(so[i + 1] - so[i]))
                    break
            nd += 1
            if nd > 5: break
    print("differing lines (first few)", nd)
