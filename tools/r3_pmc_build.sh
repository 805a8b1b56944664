#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for t in 1 0; do
  KT_BENCH_EXPORT_TARGET=$t tools/pmc_ctr.sh ext$t --workload ctr_k31 --steps 2 --warmup 1 > gpurun_out/r3_pmc_ext$t.txt 2>&1
  sed -n '/== build_kernel/,/SQ_WAVE_CYCLES/p' gpurun_out/r3_pmc_ext$t.txt
done
