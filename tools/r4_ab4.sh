#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for r in 1 2; do
tools/ab_kernels.sh "base swt1024" "--workload ctr_k31 --steps 5 --warmup 2" "part2_swwc" 2>&1 | grep -v "^$"
tools/ab_kernels.sh "base swt1024" "--workload ctr_k15 --steps 5 --warmup 2" "part2_swwc" 2>&1 | grep -v "^$"
done
