#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "export_target or cfg3 or cfg4 or two_ranks" 2>&1 | grep -E "passed|failed|rror" | head -3
tools/ab_kernels.sh "base" "--workload ctr_k31 --steps 5 --warmup 1" "part2|scatter1w|build_kernel|ext_"
tools/ab_kernels.sh "base" "--workload ctr_k15 --steps 5 --warmup 1" "part2|scatter1w|build_kernel|ext_"
