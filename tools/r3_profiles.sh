#!/bin/bash
# round 3: the rocprofv3 summaries that go to profiles/ (kernel stats + FETCH/WRITE passes per workload, the headline run)
cd "$GRAFT_REPO_ROOT" || exit 1
tools/profile_bench.sh r3_ctr_k31 --workload ctr_k31 --steps 5 --warmup 2 > /dev/null 2>&1
tools/profile_bench.sh r3_ctr_k15 --workload ctr_k15 --steps 5 --warmup 2 > /dev/null 2>&1
tools/profile_bench.sh r3_comp_cgr_k7 --workload comp_cgr_k7 --steps 5 --warmup 2 --no-place > /dev/null 2>&1
tools/profile_bench.sh r3_comp_oligo_k4 --workload comp_oligo_k4 --steps 20 --warmup 5 --no-place > /dev/null 2>&1
KT_SHARD_FORCE=1 tools/profile_bench.sh r3_ctr_k31_forced --workload ctr_k31 --steps 5 --warmup 2 > /dev/null 2>&1
tools/profile_headline.sh > /dev/null 2>&1
tools/pmc.sh r3_oligo oligo 3 > gpurun_out/r3_oligo_pmc.txt 2>&1
tools/pmc_ctr.sh r3 --workload ctr_k31 --steps 2 --warmup 1 > gpurun_out/r3_ctr_k31_pmc.txt 2>&1
ls gpurun_out/prof_r3_*/summary.txt
