#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "oligo or cfg2 or cfg5" 2>&1 | tail -3
for rep in 1 2; do
tools/ab_kernels.sh "base lut0" "--workload comp_oligo_k4 --steps 30 --warmup 5" "oligo_sb"
done
