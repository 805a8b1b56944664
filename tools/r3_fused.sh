#!/bin/bash
# round 3, first GPU call: the export-target tests, the ctr tests, ctr_k31 / ctr_k15 with and without the fused export
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "export_target or dense_table or ctr_vs_oracle or bulk_build_matches or rebuild" > gpurun_out/r3_fused_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_fused_tests.log
tail -5 gpurun_out/r3_fused_tests.log
for t in 1 0; do
  KT_BENCH_EXPORT_TARGET=$t timeout 600 python bench.py --workload ctr_k31 --steps 10 --warmup 2 --no-cpu > gpurun_out/r3_ctr31_fused$t.json 2> gpurun_out/r3_ctr31_fused$t.err
  KT_BENCH_EXPORT_TARGET=$t timeout 600 python bench.py --workload ctr_k15 --steps 10 --warmup 2 --no-cpu > gpurun_out/r3_ctr15_fused$t.json 2> gpurun_out/r3_ctr15_fused$t.err
done
python tools/show_bench.py gpurun_out/r3_ctr31_fused1.json gpurun_out/r3_ctr31_fused0.json gpurun_out/r3_ctr15_fused1.json gpurun_out/r3_ctr15_fused0.json 2>/dev/null || cat gpurun_out/r3_ctr31_fused*.json
