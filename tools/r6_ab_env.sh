#!/bin/bash
# round 6: A/B of an environment knob on one bench workload under rocprofv3 kernel stats.
# usage (GPU box): tools/r6_ab_env.sh <workload> <KEY=VAL> [more bench args]: runs without and with the variable
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
wl=$1; kv=$2; shift 2
for mode in off on; do
  out=gpurun_out/r6/abenv_${wl}_$mode; rm -rf $out; mkdir -p $out
  ( [ $mode = on ] && export "$kv"
    rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu "$@" > $out/bench.json 2> $out/bench.err )
  python3 - $out "$mode $kv" <<'PY'
import csv, sys, json
out, tag = sys.argv[1], sys.argv[2]
try:
    j = json.loads([ln for ln in open(out + "/bench.json").read().splitlines() if ln.startswith("{")][-1])
    print("%s: ms_per_step %.3f value %.2f frac %.4f check %s" % (tag, j["ms_per_step"], j["value"], j["roofline"]["frac"], j["output_check"]["ok"]))
except Exception as e:
    print(tag, "no bench line:", e)
for r in list(csv.DictReader(open(out + "/kt_kernel_stats.csv")))[:6]:
    print("   %-70s calls %4s avg %9.3f ms" % (r["Name"].replace("(anonymous namespace)::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
