#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kmers or ctr or cov or route or min or sharded" 2>&1 | tail -3
tools/ab_kernels.sh "base" "--workload ctr_k31 --steps 4 --warmup 1" "part2|scatter1w|build_kernel"
tools/ab_kernels.sh "base" "--workload ctr_k15 --steps 4 --warmup 1" "part2|scatter1w|build_kernel"
tools/ab_kernels.sh "base" "--workload cov_k15 --steps 4 --warmup 1" "cov_kernel"
