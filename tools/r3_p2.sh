#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ctr and (bulk or rebuild or dense or vs_oracle or high_load or export_target)" 2>&1 | tail -3
for rep in 1 2; do
tools/ab_kernels.sh "base p2l0" "--workload ctr_k31 --steps 4 --warmup 1" "part2|scatter1w|build_kernel"
done
tools/ab_kernels.sh "base" "--workload ctr_k15 --steps 4 --warmup 1" "part2|scatter1w|build_kernel"
