#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ctr or sharded or two_ranks" 2>&1 | grep -E "passed|failed|rror" | head -5
tools/ab_kernels.sh "base" "--workload ctr_k31 --steps 4 --warmup 1" "part2|scatter1w|build_kernel"
tools/ab_kernels.sh "base" "--workload ctr_k15 --steps 4 --warmup 1" "part2|scatter1w|build_kernel"
timeout 300 python tools/fuzz_big.py 150 77 > gpurun_out/r3_fuzz_big.txt 2>&1; grep -E "fuzz_big|Error|assert" gpurun_out/r3_fuzz_big.txt | tail -3
timeout 300 python tools/fuzz_ctr.py 120 9 > gpurun_out/r3_fuzz_ctr.txt 2>&1; grep -E "fuzz|Error|assert|ok" gpurun_out/r3_fuzz_ctr.txt | tail -3
