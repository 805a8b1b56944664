#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_bench.sh <tag> [bench.py args...]
# 1. rocprofv3 --kernel-trace --stats of the bench command   -> gpurun_out/prof_<tag>/kt_*
# 2. separate --pmc passes (FETCH_SIZE / WRITE_SIZE, no tracing domains) -> gpurun_out/prof_<tag>/pmc*
# 3. one-file text summary gpurun_out/prof_<tag>/summary.txt  (copy into profiles/ to commit)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/prof_$tag; mkdir -p $out
args="$@ --no-cpu"
rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 bench.py $args > $out/bench_under_trace.json 2> $out/kt.err
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $out/pmc1 -o p --output-format csv -- python3 bench.py $args > $out/pmc1.json 2> $out/pmc1.err
rocprofv3 --pmc WRITE_SIZE -d $out/pmc2 -o p --output-format csv -- python3 bench.py $args > $out/pmc2.json 2> $out/pmc2.err
{
  echo "# rocprofv3 summary for: python3 bench.py $args   (tag $tag)"
  echo "## bench line under --kernel-trace"; cat $out/bench_under_trace.json
  echo "## kernel stats (rocprofv3 --kernel-trace --stats)"; cat $out/kt_kernel_stats.csv
  echo "## PMC (separate passes), per-dispatch means"
  python3 tools/pmc_summary.py $out
} > $out/summary.txt 2>&1
cat $out/summary.txt | cut -c1-400
