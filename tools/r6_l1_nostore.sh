#!/bin/bash
# round 6: what level 1 costs without its copy-out (timing build, KT_BUILD_DBG=0x1000: scatter1y's stores skipped - the counts
# are wrong, only the kernel's duration means anything).  usage (GPU box): tools/r6_l1_nostore.sh   (needs variants/libabl.so:
# tools/build_variant_tu.sh abl kt_bulk -DKT_ABLATION=1)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
export KT_LIB=$GRAFT_REPO_ROOT/kmertools_amd/variants/libabl.so
for k in 31 15; do
  for dbg in 0 4096; do
    out=gpurun_out/r6/nostore_${k}_$dbg; rm -rf $out; mkdir -p $out
    ( export KT_BUILD_DBG=$dbg
      timeout 240 rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 tools/l1_phases.py $k 25000000 > $out/log.txt 2>&1 )
    python3 - $out "k=$k dbg=$dbg" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1] + "/kt_kernel_stats.csv")):
    if "scatter1y" in r["Name"] or "pack_segments" in r["Name"]:
        print("%-14s %-60s calls %3s avg %8.3f ms" % (sys.argv[2], r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
  done
done
