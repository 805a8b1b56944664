#!/bin/bash
# GPU box: per-kernel ms of the ctr step under environment variations: tools/r5_env_sweep.sh "<VAR=val ...>" "<VAR=val ...>" ... [-- run args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done
[ "$1" = "--" ] && shift
for e in "${sets[@]}"; do
  out=gpurun_out/sweep; rm -rf $out; mkdir -p $out
  env $e timeout 300 rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 tools/r5_ctr_run.py "$@" > $out/out.txt 2> $out/err.txt
  echo "== [$e]  $(tail -1 $out/out.txt)"
  python3 - "$out/kt_kernel_stats.csv" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search("build_kernel|scatter1|part2_swwc", r["Name"]):
        n = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Name"])[:40]
        print("   %-42s avg %8.3f ms" % (n, float(r["AverageNs"]) / 1e6))
PY
done
