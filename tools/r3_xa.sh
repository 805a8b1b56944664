#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tools/ab_kernels.sh "xa4" "--workload ctr_k31 --steps 4 --warmup 1" "build_kernel|ext_"
KT_BUILD_WGS=16 KT_BENCH_EXPORT_TARGET=0 tools/ab_kernels.sh "base" "--workload ctr_k31 --steps 4 --warmup 1" "build_kernel|ext_"
