#!/bin/bash
# GPU box: the ctr / shard parity tests, then the per-kernel times of the ctr step (k=31, k=15) -> gpurun_out/try.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out/try.log; : > $O
[ "$1" = notest ] || timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "ctr or bulk or shard or route or smoke" 2>&1 | grep -E "passed|failed|error" | tail -3 >> $O
cd /tmp && export TMPDIR=/tmp
for k in 31 15; do
  R=25000000; [ $k = 15 ] && R=50000000
  rm -rf /tmp/p$k
  rocprofv3 --kernel-trace --stats -d /tmp/p$k -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/r5_ctr_run.py --k $k --reads $R --steps 4 2>/dev/null | tail -1 >> $O
  f=$(find /tmp/p$k -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $O <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    for t in ("build_kernel","scatter1y","scatter1x","part2_swwc"):
        if t in n: print(t, r["Calls"], "avg ms %.3f"%(float(r["AverageNs"])/1e6), "min %.3f"%(float(r["MinNs"])/1e6))
P
done
cat $O
