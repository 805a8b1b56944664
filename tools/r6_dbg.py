"""debug: which k-mers does the routed path lose?  one random read, k=31, positions of the lost k-mers modulo the tile shapes"""
import os, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np
os.environ["KT_SHARD_FORCE"] = sys.argv[1] if len(sys.argv) > 1 else "4"
os.environ["KT_BULK_MIN_BASES"] = "0"
from kmertools_amd import device
from oracle import kt_oracle as oracle
k = 31
rng = np.random.default_rng(3)
L = 60000
seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].tobytes()
bases, offsets = device.to_csr([seq])
ctx = device.Context(0)
sh = device.Sharded(ctx, k, 1 << 21, L, 1, 0, None)
sh.add_reads_host(bases, offsets)
sh.finalize()
gk, gc = sh.table.export_host()
f, r, _ = oracle.kmers(seq, k)
canon = np.minimum(f, r)
wk = np.unique(canon)
print("got", len(gk), "want", len(wk), "sum", gc.sum(), "want", len(canon))
lost = np.setdiff1d(wk, gk)
pos = np.flatnonzero(np.isin(canon, lost))
print("lost positions:", pos[:40])
print("mod 32:", np.unique(pos % 32, return_counts=True))
print("mod 2048 // 32:", np.unique((pos % 2048) // 32, return_counts=True))
dup = gk[gc > 1]
print("counts>1:", len(dup))
