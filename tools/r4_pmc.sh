#!/bin/bash
# GPU box: SQ / LDS counters of the ctr k=31 kernels with level 1 = scatter1c and = scatter1w
cd "$GRAFT_REPO_ROOT" || exit 1
tools/pmc_ctr.sh c1 --workload ctr_k31 --steps 2 --warmup 1 > gpurun_out/r4_pmc_c1.txt 2>&1
KT_S1_COMB=0 tools/pmc_ctr.sh w1 --workload ctr_k31 --steps 2 --warmup 1 > gpurun_out/r4_pmc_w1.txt 2>&1
grep -A17 "== scatter1c\|== part2_swwc\|== build_kernel" gpurun_out/r4_pmc_c1.txt
grep -A17 "== scatter1w" gpurun_out/r4_pmc_w1.txt
