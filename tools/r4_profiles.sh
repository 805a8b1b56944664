#!/bin/bash
# round 4: the rocprofv3 summaries that go to profiles/ (kernel stats + FETCH/WRITE passes per workload, the headline run)
cd "$GRAFT_REPO_ROOT" || exit 1
tools/profile_bench.sh r4_ctr_k31 --workload ctr_k31 --steps 5 --warmup 2 > /dev/null 2>&1
tools/profile_bench.sh r4_ctr_k31_genome --workload ctr_k31 --genome 1000000000 --steps 3 --warmup 1 > /dev/null 2>&1
tools/profile_bench.sh r4_ctr_k15 --workload ctr_k15 --steps 5 --warmup 2 > /dev/null 2>&1
tools/profile_bench.sh r4_comp_cgr_k7 --workload comp_cgr_k7 --steps 5 --warmup 2 > /dev/null 2>&1
tools/profile_bench.sh r4_comp_oligo_k4 --workload comp_oligo_k4 --steps 20 --warmup 5 > /dev/null 2>&1
KT_SHARD_FORCE=1 tools/profile_bench.sh r4_ctr_k31_forced --workload ctr_k31 --steps 5 --warmup 2 > /dev/null 2>&1
KT_SHARD_FORCE=1 KT_BULK_MAX_B2=9 tools/profile_bench.sh r4_ctr_k31_forced_presplit --workload ctr_k31 --steps 5 --warmup 2 > /dev/null 2>&1
tools/profile_headline.sh > /dev/null 2>&1
tools/pmc_ctr.sh r4 --workload ctr_k31 --steps 2 --warmup 1 > gpurun_out/r4_ctr_k31_pmc.txt 2>&1
tools/ubench/scatter_runs 256 > gpurun_out/r4_scatter_runs_ubench.txt 2>&1
tools/ubench/scatter_runs 256 fine >> gpurun_out/r4_scatter_runs_ubench.txt 2>&1
tools/ubench/scatter_runs 128 >> gpurun_out/r4_scatter_runs_ubench.txt 2>&1
ls gpurun_out/prof_r4_*/summary.txt
