#!/bin/bash
# usage (GPU box): tools/kind_and_ctr.sh  -> which kind of process / box is this (oligo k=4 with 32 and 96 workgroups per
# slot), and do the ctr kernels care (level-1 workgroup count, build workgroup count)?
cd "$GRAFT_REPO_ROOT"
python3 tools/oligo_prio_test.py 2>/dev/null | grep "debug  0"
for m in 1 8; do echo "KT_BULK_G_MULT=$m"; KT_BULK_G_MULT=$m tools/ab_kernels.sh base "--workload ctr_k31 --steps 3 --warmup 1" "scatter1w|part2|build|export" | grep -E "ms_per_step|avg"; done
for w in 8 64 512; do echo "KT_BUILD_WGS=$w"; KT_BUILD_WGS=$w tools/ab_kernels.sh base "--workload ctr_k31 --steps 3 --warmup 1" "build" | grep -E "avg"; done
