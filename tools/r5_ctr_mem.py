"""Per-pass host-memory log of `kmertools ctr` out of core: the failing case's input (700 K x 150 bp,
~84 M distinct 31-mers) counted in >= 4 and >= 16 passes with KT_CLI_TIMING=1; stderr of both runs goes to the file named on
the command line.  Extra environment (e.g. MALLOC_ARENA_MAX=2) is taken from KT_MEM_EXTRA_ENV="A=1,B=2"."""
import os
import pathlib
import subprocess
import sys
import tempfile

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
CLI = ROOT / "kmertools_amd" / "bin" / "kmertools"


def main():
    out_path = pathlib.Path(sys.argv[1])
    passes_list = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4,16").split(",")]
    n, L, k = 700_000, 150, 31
    rng = np.random.default_rng(31337)
    codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="r5mem"))
    fa = tmp / "big.fa"
    with open(fa, "wb") as f:
        for i in range(n):
            f.write(b">r%d\n" % i)
            f.write(letters[i].tobytes())
            f.write(b"\n")
    distinct = n * (L - k + 1)  # uniform 31-mers: nearly all distinct
    extra = dict(kv.split("=", 1) for kv in os.environ.get("KT_MEM_EXTRA_ENV", "").split(",") if kv)
    env = dict(os.environ, KT_CLI_TIMING="1", KT_CLI_BATCH_BASES=str(8 << 20), **extra)
    with open(out_path, "w") as log:
        for passes in passes_list:
            small = max(1024, int(1.4 * distinct / passes))
            out = tmp / ("p%d" % passes)
            r = subprocess.run([str(CLI), "ctr", "-i", str(fa), "-o", str(out), "-k", str(k), "-m", "6"],
                               env=dict(env, KT_CTR_MAX_SLOTS=str(small)), capture_output=True, text=True)
            log.write("==== passes >= %d (KT_CTR_MAX_SLOTS=%d) rc=%d extra=%s\n" % (passes, small, r.returncode, extra))
            log.write(r.stderr)
            log.write("lines: %s\n" % subprocess.run(["wc", "-l", str(out / "kmers.counts")], capture_output=True, text=True).stdout)
            (out / "kmers.counts").unlink(missing_ok=True)
    print(open(out_path).read())


if __name__ == "__main__":
    main()
