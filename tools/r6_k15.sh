#!/bin/bash
# round 6: ctr k=15 (cfg3) under rocprofv3 kernel stats; usage (GPU box): tools/r6_k15.sh <tag> [env KEY=VAL ...]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
for kv in "$@"; do export "$kv"; done
out=gpurun_out/r6/prof_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 bench.py --workload ctr_k15 --steps 5 --warmup 2 --no-cpu > $out/bench.json 2> $out/bench.err
python3 - $out <<'PY'
import csv, sys, json
out = sys.argv[1]
try:
    j = json.loads([ln for ln in open(out + "/bench.json").read().splitlines() if ln.startswith("{")][-1])
    print("ms_per_step %.3f value %.2f frac %.4f" % (j["ms_per_step"], j["value"], j["roofline"]["frac"]))
except Exception as e:
    print("no bench line:", e)
for r in list(csv.DictReader(open(out + "/kt_kernel_stats.csv")))[:7]:
    print("%-70s calls %4s avg %9.3f ms" % (r["Name"].replace("(anonymous namespace)::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
