"""CPU oracle, ctr k=31: in-memory counting rate against the number of threads (run on the GPU box's host cores).
usage: python tools/cpu_ctr_scaling.py [reads]"""
import sys
import time
sys.path.insert(0, ".")
from oracle import kt_oracle as o

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
hb, ho = o.synth_reads(0x6b6d6572, n, 150)
for T in (1, 4, 16, 32):
    m = n if T >= 16 else n // (16 // T)          # bounded time: fewer reads for fewer threads, same table size
    c = o.Counter(max(T, 16))
    c.reserve(n * 120)
    t0 = time.perf_counter()
    c.add_reads(hb[:m * 150], ho[:m + 1], 31, threads=T)
    dt = time.perf_counter() - t0
    print("T=%2d: %8d reads in %6.2f s = %.4f Gbases/s (%d distinct)" % (T, m, dt, m * 150 / dt / 1e9, c.size()), flush=True)
    del c
