#!/bin/bash
# usage (GPU box): tools/r4_abl.sh <lib name|base> "<KT_BUILD_DBG values>" [r5_ctr_run.py args]: per-kernel ms per value
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
v=$1; vals=$2; shift 2
lib=$PWD/kmertools_amd/variants/lib$v.so; [ $v = base ] && lib=$PWD/kmertools_amd/libkmertools_hip.so
for d in $vals; do
  out=gpurun_out/abl_${v}_$d; rm -rf $out; mkdir -p $out
  KT_LIB=$lib KT_BUILD_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 tools/r5_ctr_run.py "$@" > $out/out.txt 2> $out/err.txt
  echo "== $v dbg=$d  $(tail -1 $out/out.txt)"
  python3 - "$out/kt_kernel_stats.csv" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search("build_kernel|scatter1|part2", r["Name"]):
        n = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Name"])[:40]
        print("   %-42s avg %8.3f ms" % (n, float(r["AverageNs"]) / 1e6))
PY
done
