#!/bin/bash
# round 6: the rocprofv3 summaries that go to profiles/ (kernel stats + FETCH_SIZE / WRITE_SIZE passes per workload - the two
# oligo kernels among them -, the headline run, the forced one-GPU run of the 8-rank path, the SQ / LDS counters of the ctr
# kernels).  usage (GPU box, repo root): tools/r6_profiles.sh [which...]   (default: all)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
want=" ${*:-k31 genome k15 forced8 forced8g oligo cgr headline sq} "
has() { [[ "$want" == *" $1 "* ]]; }
has k31 && tools/profile_bench.sh r6_ctr_k31 --workload ctr_k31 --steps 5 --warmup 2 > /dev/null 2>&1
has genome && tools/profile_bench.sh r6_ctr_k31_genome --workload ctr_k31 --genome 1000000000 --steps 3 --warmup 1 > /dev/null 2>&1
has k15 && tools/profile_bench.sh r6_ctr_k15 --workload ctr_k15 --steps 5 --warmup 2 > /dev/null 2>&1
has forced8 && KT_SHARD_FORCE=8 tools/profile_bench.sh r6_ctr_k31_forced8 --workload ctr_k31 --steps 5 --warmup 2 > /dev/null 2>&1
has forced8g && KT_SHARD_FORCE=8 tools/profile_bench.sh r6_ctr_k31_forced8_genome --workload ctr_k31 --genome 1000000000 --steps 3 --warmup 1 > /dev/null 2>&1
has oligo && tools/profile_bench.sh r6_comp_oligo_k4 --workload comp_oligo_k4 --steps 20 --warmup 5 > /dev/null 2>&1
has cgr && tools/profile_bench.sh r6_comp_cgr_k7 --workload comp_cgr_k7 --steps 5 --warmup 2 > /dev/null 2>&1
has headline && tools/profile_headline.sh r6 > /dev/null 2>&1
has sq && tools/pmc_ctr.sh r6 --workload ctr_k31 --steps 2 --warmup 1 > gpurun_out/r6_ctr_k31_pmc.txt 2>&1
ls gpurun_out/prof_r6_*/summary.txt
