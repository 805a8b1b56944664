#!/bin/bash
# per-kernel times of the fused and the unfused ctr step, the export-target tests again, the cursor micro-benchmark
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "export_target" > gpurun_out/r3_fused_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_fused_tests.log
tail -5 gpurun_out/r3_fused_tests.log
timeout 300 tools/ubench/same_addr_atomic > gpurun_out/r3_same_addr_atomic.txt 2>&1
cat gpurun_out/r3_same_addr_atomic.txt
for t in 1 0; do
  echo "### KT_BENCH_EXPORT_TARGET=$t"
  KT_BENCH_EXPORT_TARGET=$t tools/ab_kernels.sh base "--workload ctr_k31 --steps 5 --warmup 1" "build_kernel|scatter1|part2|export"
  mv gpurun_out/abk_base gpurun_out/abk_fused$t
done
