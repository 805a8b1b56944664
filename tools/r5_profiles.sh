#!/bin/bash
# round 5: the rocprofv3 summaries that go to profiles/ (kernel stats + FETCH/WRITE passes per workload, the headline run,
# the SQ / LDS counters of the ctr kernels, the micro-benchmarks of the round)
cd "$GRAFT_REPO_ROOT" || exit 1
sed -i 's/prof_r4_headline/prof_r5_headline/' tools/profile_headline.sh
tools/profile_bench.sh r5_ctr_k31 --workload ctr_k31 --steps 5 --warmup 2 > /dev/null 2>&1
tools/profile_bench.sh r5_ctr_k31_genome --workload ctr_k31 --genome 1000000000 --steps 3 --warmup 1 > /dev/null 2>&1
tools/profile_bench.sh r5_ctr_k15 --workload ctr_k15 --steps 5 --warmup 2 > /dev/null 2>&1
KT_SHARD_FORCE=1 tools/profile_bench.sh r5_ctr_k31_forced --workload ctr_k31 --steps 5 --warmup 2 > /dev/null 2>&1
KT_SHARD_FORCE=1 KT_BULK_MAX_B2=9 tools/profile_bench.sh r5_ctr_k31_forced_presplit --workload ctr_k31 --steps 5 --warmup 2 > /dev/null 2>&1
tools/profile_headline.sh > /dev/null 2>&1
tools/pmc_ctr.sh r5 --workload ctr_k31 --steps 2 --warmup 1 > gpurun_out/r5_ctr_k31_pmc.txt 2>&1
(cd tools/ubench && ./xcd_append 256) > gpurun_out/r5_xcd_append_ubench.txt 2>&1
(cd tools/ubench && ./store_overlap) > gpurun_out/r5_store_overlap_ubench.txt 2>&1
ls gpurun_out/prof_r5_*/summary.txt
