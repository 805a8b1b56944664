#!/bin/bash
# usage (GPU box): tools/env_ab.sh "<VAR=a VAR=b ...>" "<workloads>" [reps] [extra bench args]  - kernel_ms per env setting
sets=$1; wls=$2; reps=${3:-2}; shift 3
for r in $(seq $reps); do for w in $wls; do for e in $sets; do
  ms=$(env $e timeout 120 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu "$@" 2>&1 | grep -o "kernel_ms\": [0-9.]*" | cut -d' ' -f2)
  echo "$w $e $ms"
done; done; done
