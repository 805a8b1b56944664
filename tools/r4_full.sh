#!/bin/bash
# the whole GPU test suite + the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gpu_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_gpu_tests.log
tail -6 gpurun_out/r4_gpu_tests.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
echo "bench rc=$?"; tail -3 gpurun_out/r4_bench_default.err
python tools/show_bench.py gpurun_out/r4_bench_default.json
