#!/bin/bash
# GPU box: ctr k=31 / k=15 per-kernel times against the requested table capacity (range geometry: fewer, fuller ranges)
cd "$GRAFT_REPO_ROOT" || exit 1
for f in 1.9 1.6 1.43 1.36; do
  echo "#### KT_BENCH_CAP_FACTOR=$f ctr_k31"
  KT_BENCH_CAP_FACTOR=$f tools/ab_kernels.sh base "--workload ctr_k31 --steps 5 --warmup 2" 2>&1 | tail -8
  grep -o '"table_slots_rank0": [0-9]*' gpurun_out/abk_base/bench.json
done
for f in 1.9 1.5; do
  echo "#### KT_BENCH_CAP_FACTOR=$f ctr_k15"
  KT_BENCH_CAP_FACTOR=$f tools/ab_kernels.sh base "--workload ctr_k15 --steps 5 --warmup 2" 2>&1 | tail -8
  grep -o '"table_slots_rank0": [0-9]*' gpurun_out/abk_base/bench.json
done
