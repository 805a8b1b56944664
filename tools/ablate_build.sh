#!/bin/bash
# usage (GPU box): tools/ablate_build.sh "<dbg values>" [bench args]: build_kernel time per KT_BUILD_DBG value
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
vals=$1; shift
for v in $vals; do
  out=gpurun_out/abl_$v; mkdir -p $out
  KT_BUILD_DBG=$v rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 bench.py "$@" --no-cpu --no-export > $out/bench.json 2> $out/err.txt
  echo "dbg=$v $(grep build_kernel $out/kt_kernel_stats.csv | awk -F, '{printf "build avg %.3f ms", $(NF-4)/1e6}')"
done
