#!/bin/bash
# GPU box: the sliced (multi-GPU) path forced on one GPU: per-kernel times, with level 2 restricted so that the pre-split runs
cd "$GRAFT_REPO_ROOT" || exit 1
for mb in 10 9 8; do
  echo "### KT_SHARD_FORCE=1 KT_BULK_MAX_B2=$mb"
  KT_SHARD_FORCE=1 KT_BULK_MAX_B2=$mb tools/ab_kernels.sh base "--workload ctr_k31 --steps 4 --warmup 1" "build_kernel|scatter1|part2|ext_|page_tails"
done
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharded or two_ranks or bench_two_rank" 2>&1 | tail -3
