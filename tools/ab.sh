#!/bin/bash
# usage (on the GPU box): tools/ab.sh "A B C" [reps]   with RS/OV env for sweep_oligo.py
# interleaves the variants rep times and prints the min / median ms per (variant, R, oversub)
vars=$1; reps=${2:-3}
tmp=$(mktemp)
for rep in $(seq $reps); do for v in $vars; do
  KT_LIB=$PWD/kmertools_amd/variants/lib$v.so timeout 300 python tools/sweep_oligo.py 2>&1 | grep "R=" | sed "s/^/$v /" >> $tmp
done; done
python3 - $tmp <<'PY'
import sys, collections, statistics
d = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    p = ln.split()
    d[(p[0], p[2], p[3])].append(float(p[4]))
for k in sorted(d):
    v = d[k]
    print("%-6s R=%-3s %-10s min %.3f ms  median %.3f ms  (n=%d)  -> %.0f Gbases/s" % (k[0], k[1], k[2], min(v), statistics.median(v), len(v), 1.5e3 / min(v)))
PY
