#!/bin/bash
# usage (GPU box): tools/kstats.sh <tag> <python script + args...>   -> per-kernel average times
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
tag=$1; shift
[ $# -ge 1 ] || { echo "usage: tools/kstats.sh <tag> <script> [args]"; exit 2; }
out=gpurun_out/ks_$tag; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 "$@" > $out/stdout.txt 2> $out/stderr.txt
python3 - $out <<'PY'
import csv, sys
out = sys.argv[1]
print(open(out + "/stdout.txt").read()[:600])
for r in csv.DictReader(open(out + "/kt_kernel_stats.csv")):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
    print("%-28s calls %4s  avg %9.3f ms  %5.1f%%" % (n, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
PY
