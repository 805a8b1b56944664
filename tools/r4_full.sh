#!/bin/bash
# the whole GPU test suite + the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
# (faulthandler: a test stuck in a C call - a device-wide synchronize that never returns - prints where after 150 s)
timeout 700 python -m pytest tests -m gpu -x -q -o faulthandler_timeout=150 > gpurun_out/r4_gpu_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_gpu_tests.log
tail -6 gpurun_out/r4_gpu_tests.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
echo "bench rc=$?"; tail -3 gpurun_out/r4_bench_default.err
python tools/show_bench.py gpurun_out/r4_bench_default.json
