#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_cli.py -x -q -m gpu -k "sharded or two_ranks or out_of_core or bench_two_rank or export_target or cfg4" > gpurun_out/r3_shard_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_shard_tests.log
tail -25 gpurun_out/r3_shard_tests.log
KT_SHARD_FORCE=1 timeout 600 python bench.py --workload ctr_k31 --steps 5 --warmup 1 --no-cpu > gpurun_out/r3_ctr31_forced.json 2> gpurun_out/r3_ctr31_forced.err; tail -3 gpurun_out/r3_ctr31_forced.err; python tools/show_bench.py gpurun_out/r3_ctr31_forced.json
