#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for w in 8 16 32 64; do echo "## KT_BUILD_WGS_EXT=$w"; KT_BUILD_WGS_EXT=$w tools/ab_kernels.sh "base" "--workload ctr_k31 --steps 4 --warmup 1" "build_kernel|ext_"; done
