#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python tools/r4_dbg1.py 2>&1 | tail -4
for r in 1 2; do
tools/ab_kernels.sh "base c512" "--workload ctr_k31 --steps 5 --warmup 2" "scatter1|part2_swwc|build" 2>&1 | grep -v "^$"
KT_S1_COMB=0 tools/ab_kernels.sh "base" "--workload ctr_k31 --steps 5 --warmup 2" "scatter1|part2_swwc|build" 2>&1 | grep -v "^$"
done
tools/ab_kernels.sh "base c512" "--workload ctr_k15 --steps 5 --warmup 2" "scatter1|part2_swwc|build" 2>&1 | grep -v "^$"
KT_S1_COMB=0 tools/ab_kernels.sh "base" "--workload ctr_k15 --steps 5 --warmup 2" "scatter1|part2_swwc|build" 2>&1 | grep -v "^$"
