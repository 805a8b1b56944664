#!/bin/bash
# round 6: A/B of library builds (kmertools_amd/variants/lib<name>.so, tools/build_variant_tu.sh; "main" = the shipped one) on one
# bench workload under rocprofv3 kernel stats, all on the same box.  usage (GPU box): tools/r6_ab_libs.sh <workload> <name>...
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
wl=$1; shift
for name in "$@"; do
  out=gpurun_out/r6/ablib_${wl}_$name; rm -rf $out; mkdir -p $out
  ( [ $name != main ] && export KT_LIB=$GRAFT_REPO_ROOT/kmertools_amd/variants/lib$name.so
    timeout 300 rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 bench.py --workload $wl --steps 6 --warmup 2 --no-cpu > $out/bench.json 2> $out/bench.err )
  python3 - $out "$name" <<'PY'
import csv, sys, json
out, tag = sys.argv[1], sys.argv[2]
try:
    j = json.loads([ln for ln in open(out + "/bench.json").read().splitlines() if ln.startswith("{")][-1])
    print("%-10s ms_per_step %.3f value %.2f check %s" % (tag, j["ms_per_step"], j["value"], j["output_check"]["ok"]), end="")
except Exception as e:
    print(tag, "no bench line:", e, end="")
for r in csv.DictReader(open(out + "/kt_kernel_stats.csv")):
    n = r["Name"]
    for key in ("build_kernel", "scatter1y", "part2_swwc", "pack_segments"):
        if key in n:
            print("  %s %.3f" % (key, float(r["AverageNs"]) / 1e6), end="")
print()
PY
done
