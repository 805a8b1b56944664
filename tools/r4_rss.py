import os, subprocess, sys, pathlib, tempfile
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from oracle import kt_oracle as oracle
n, L = 350000, 150
hb, ho = oracle.synth_reads(31337, n, L, noise=False)
d = pathlib.Path(tempfile.mkdtemp())
fa = d / "big.fa"
with open(fa, "wb") as f:
    for i in range(n):
        f.write(b">r%d\n" % i); f.write(hb[int(ho[i]):int(ho[i+1])].tobytes()); f.write(b"\n")
cli = str(pathlib.Path(__file__).resolve().parent.parent / "kmertools_amd/bin/kmertools")
for env in ({"KT_CTR_MAX_SLOTS": "15000000"}, {}, {"KT_CTR_MAX_SLOTS": "15000000", "KT_BULK": "0"}):
    e = dict(os.environ, KT_CLI_TIMING="1", KT_CLI_BATCH_BASES=str(8 << 20)); e.update(env)
    r = subprocess.run([cli, "ctr", "-i", str(fa), "-o", str(d / "o"), "-k", "31"], env=e, capture_output=True, text=True)
    print(env, r.returncode); print("\n".join(l for l in r.stderr.splitlines() if "VmHWM" in l or "pass(es)" in l))
