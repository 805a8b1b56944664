#!/bin/bash
# usage (GPU box): [ENVVARS] tools/ab_kernels.sh "<variant libs|base>" "<bench args>" [kernel-name-regex]
# per variant: rocprofv3 kernel stats of one bench run, the matching kernels' average ms
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
vars=$1; bargs=$2; pat=${3:-build_kernel|scatter1|part2|export}
for v in $vars; do
  out=gpurun_out/abk_$v; rm -rf $out; mkdir -p $out
  lib=$PWD/kmertools_amd/variants/lib$v.so; [ $v = base ] && lib=$PWD/kmertools_amd/libkmertools_hip.so
  KT_LIB=$lib rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 bench.py $bargs --no-cpu > $out/bench.json 2> $out/err.txt
  echo "== $v  $(grep -o '"ms_per_step": [0-9.]*' $out/bench.json | head -1)"
  python3 - "$out/kt_kernel_stats.csv" "$pat" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        n = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Name"])[:64]
        print("   %-66s avg %8.3f ms" % (n, float(r["AverageNs"]) / 1e6))
PY
done
