#!/bin/bash
# round 6: A/B of route_kernel variants (KT_LIB builds, KT_ROUTE_GRID) - the forced 8-owner ctr_k31 step, 12 M reads, under
# rocprofv3 kernel stats: prints the route kernel's average per variant.  usage (GPU box): tools/r6_route_ab.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
run() {  # tag, env...
  tag=$1; shift
  out=gpurun_out/r6/ab_$tag; mkdir -p $out
  ( export KT_SHARD_FORCE=8; for kv in "$@"; do export "$kv"; done
  rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 bench.py --workload ctr_k31 --steps 3 --warmup 1 --no-cpu --reads 12000000 > $out/bench.json 2> $out/bench.err )
  python3 - $out $tag <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1] + "/kt_kernel_stats.csv")):
    if "route_kernel" in r["Name"] or "scatter1y" in r["Name"]:
        print("%-12s %-40s calls %3s avg %8.3f ms" % (sys.argv[2], r["Name"].replace("(anonymous namespace)::", "")[:40], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
}
# NOTE: rocprofv3 must launch python3 directly; env vars are exported instead of using `env`
# (the variants of the round - waves per workgroup, ablations - were builds with -DKT_ROUTE_WAVES / -DKT_ROUTE_ABL:
# tools/build_variant_tu.sh <name> kt_shard -D..., then `run <name> KT_LIB=.../variants/lib<name>.so`)
run base
run g20 KT_ROUTE_GRID=20
run g32 KT_ROUTE_GRID=32
