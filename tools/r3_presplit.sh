#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for mb in 11 8; do
  echo "### KT_SHARD_FORCE=1 KT_BULK_MAX_B2=$mb"
  KT_SHARD_FORCE=1 KT_BULK_MAX_B2=$mb tools/ab_kernels.sh base "--workload ctr_k31 --steps 4 --warmup 1" "build_kernel|scatter1|part2|ext_|page_tails"
done
