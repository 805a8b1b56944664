#!/bin/bash
# what the driver runs at round end: the GPU suite, smoke(), the default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_gpu_tests.log
grep -E "passed|failed|rc=" gpurun_out/r3_gpu_tests.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error|rror" | tail -3
timeout 900 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo "bench rc=$?"
python tools/show_bench.py gpurun_out/r3_bench_default.json
