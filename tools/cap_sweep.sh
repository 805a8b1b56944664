#!/bin/bash
# usage (GPU box): tools/cap_sweep.sh <workload> "<cap requests>" [reps]   - kernel_ms per requested table capacity
wl=$1; caps=$2; reps=${3:-2}
for r in $(seq $reps); do for c in $caps; do
  ms=$(timeout 120 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu --cap-slots $c 2>&1 | grep -o "kernel_ms\": [0-9.]*" | cut -d' ' -f2)
  echo "$wl cap=$c $ms"
done; done
