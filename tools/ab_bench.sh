#!/bin/bash
# usage (GPU box): tools/ab_bench.sh "<variant libs>" "<workloads>" [reps] [extra bench args]
# interleaves variants x workloads and prints kernel_ms of each run
vars=$1; wls=$2; reps=${3:-2}; shift 3
for r in $(seq $reps); do for w in $wls; do for v in $vars; do
  ms=$(KT_LIB=$PWD/kmertools_amd/variants/lib$v.so timeout 120 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu "$@" 2>&1 | grep -o "kernel_ms\": [0-9.]*" | cut -d' ' -f2)
  echo "$w $v $ms"
done; done; done
