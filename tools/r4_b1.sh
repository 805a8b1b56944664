#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for b in 0 8 7; do
  echo "### k=15 KT_BULK_B1=$b"
  KT_BULK_B1=$b tools/ab_kernels.sh base "--workload ctr_k15 --steps 4 --warmup 1" "build_kernel|scatter1|part2_swwc|part2_fast"
done
for b in 0 9; do
  echo "### k=31 KT_BULK_B1=$b"
  KT_BULK_B1=$b tools/ab_kernels.sh base "--workload ctr_k31 --steps 4 --warmup 1" "build_kernel|scatter1|part2_swwc|part2_fast"
done
