#!/bin/bash
# GPU box: quick parity (ctr tests) + per-kernel times of ctr k=31 / k=15
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ctr or bulk or shard or route" > gpurun_out/r4_quick_tests.log 2>&1
tail -4 gpurun_out/r4_quick_tests.log
tools/ab_kernels.sh base "--workload ctr_k31 --steps 5 --warmup 2" 2>&1 | tail -6
tools/ab_kernels.sh base "--workload ctr_k15 --steps 5 --warmup 2" 2>&1 | tail -6
