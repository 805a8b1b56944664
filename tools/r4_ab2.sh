#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for r in 1 2; do
tools/ab_kernels.sh "base t1024" "--workload ctr_k31 --steps 5 --warmup 2" "part2" 2>&1 | grep -v "^$"
KT_P2_FAST=0 tools/ab_kernels.sh "base" "--workload ctr_k31 --steps 5 --warmup 2" "part2" 2>&1 | grep -v "^$"
done
tools/r4_abl.sh abl "0 256 512 1024 1280" 2>&1 | grep -v "scatter1w\|build_kernel"
