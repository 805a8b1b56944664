#!/bin/bash
# usage (GPU box): tools/profile_headline.sh  -> the default bench.py command (the driver's) under rocprofv3
# --kernel-trace --stats, its JSON line, and FETCH_SIZE / WRITE_SIZE of the oligo kernel from separate --pmc passes
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r3_headline; rm -rf $out; mkdir -p $out
python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu > $out/bench_under_trace.json 2> $out/kt.err
{
  echo "# python3 bench.py --steps 20 --warmup 5   (the driver's command; JSON line of the plain run)"
  cat $out/bench.json
  echo "## the same command with --no-cpu under rocprofv3 --kernel-trace --stats: JSON line, then kernel stats"
  cat $out/bench_under_trace.json
  cat $out/kt_kernel_stats.csv
} > $out/summary.txt
cut -c1-300 $out/summary.txt | head -30
