#!/bin/bash
# round 6: build_kernel's ablation switches (timing build: variants/libabl.so, KT_BUILD_DBG bit 0 = no inserts, bit 1 = no stores,
# 0x40 = every wave's run starts on a line) - the kernel's duration per switch.  usage (GPU box): tools/r6_build_abl.sh [k] [dbg...]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
export KT_LIB=$GRAFT_REPO_ROOT/kmertools_amd/variants/${KT_ABL_LIB:-libabl.so}
k=${1:-31}; shift
for dbg in ${@:-0 1}; do   # (never bit 1 with an export target: the hole scan behind the build does not end)
  out=gpurun_out/r6/babl_${k}_$dbg; rm -rf $out; mkdir -p $out
  ( export KT_BUILD_DBG=$dbg
    timeout 240 rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 tools/l1_phases.py $k 25000000 > $out/log.txt 2>&1 )
  grep "^build" $out/log.txt | cut -c1-250
  python3 - $out "k=$k dbg=$dbg" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1] + "/kt_kernel_stats.csv")):
    if "build_kernel" in r["Name"]:
        print("%-14s %-60s calls %3s avg %8.3f ms" % (sys.argv[2], r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
