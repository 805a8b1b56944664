#!/bin/bash
# usage (GPU box): tools/kstats2.sh <tag> [bench.py args...]  -> kernel-trace stats only, printed and kept in gpurun_out/ks_<tag>.csv
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/ks_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 bench.py "$@" --no-cpu > $out/bench.json 2> $out/err.txt
python3 - "$out/kt_kernel_stats.csv" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    name=r["Name"]; 
    import re
    short=re.sub(r"\(anonymous namespace\)::","",name)[:70]
    print("%-72s calls %4s avg %10.3f ms  total %9.1f ms" % (short, r["Calls"], float(r["AverageNs"])/1e6, float(r["TotalDurationNs"])/1e6))
PY
cat $out/bench.json | cut -c1-300
