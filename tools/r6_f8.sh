#!/bin/bash
# round 6: the forced one-GPU run of the 8-rank path (KT_SHARD_FORCE=8: route into 8 owners' regions, all counted here) under
# rocprofv3 --kernel-trace --stats; usage (GPU box, repo root): tools/r6_f8.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=gpurun_out/r6/prof_$tag; mkdir -p $out
KT_SHARD_FORCE=${KT_SHARD_FORCE:-8} rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 bench.py --workload ctr_k31 --steps 5 --warmup 2 --no-cpu "$@" > $out/bench.json 2> $out/bench.err
python3 - $out <<'PY'
import csv, sys, json
out = sys.argv[1]
try:
    j = json.loads([ln for ln in open(out + "/bench.json").read().splitlines() if ln.startswith("{")][-1])
    print("ms_per_step %.3f value %.2f exchanged %s" % (j["ms_per_step"], j["value"], j.get("exchanged_bytes_per_rank")))
except Exception as e:
    print("no bench line:", e)
for r in list(csv.DictReader(open(out + "/kt_kernel_stats.csv")))[:9]:
    print("%-70s calls %4s avg %9.3f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
