#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "ctr or sharded or two_ranks or cov" > gpurun_out/r3_ctr_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_ctr_tests.log
tail -8 gpurun_out/r3_ctr_tests.log
timeout 600 python bench.py --workload ctr_k31 --steps 5 --warmup 1 --no-cpu > gpurun_out/r3_ctr31.json 2> gpurun_out/r3_ctr31.err; python tools/show_bench.py gpurun_out/r3_ctr31.json
