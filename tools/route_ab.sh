#!/bin/bash
# usage (GPU box): tools/route_ab.sh [n_reads] [devices]  -> `kmertools ctr --devices N` with N rank threads sharing the
# one GPU (KT_CLI_SHARE_GPU=1, in-process host all-to-all), under rocprofv3 kernel stats, with the wide route kernel
# and with the one-lane-one-key one; the outputs must be identical
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
n=${1:-4000000}; dev=${2:-8}
out=gpurun_out/route_ab; rm -rf $out; mkdir -p $out
python3 - "$n" /tmp/route_ab.fa <<'PY'
import sys, numpy as np
n = int(sys.argv[1]); rng = np.random.default_rng(5)
b = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=(n, 150))]
rows = np.empty((n, 150 + 4), np.uint8); rows[:, 0] = ord(">"); rows[:, 1] = ord("r"); rows[:, 2] = 10; rows[:, 3:153] = b; rows[:, 153] = 10
rows.tofile(sys.argv[2])
PY
for w in 1 0; do
  KT_ROUTE_WIDE=$w KT_CLI_SHARE_GPU=1 KT_CLI_TIMING=1 rocprofv3 --kernel-trace --stats -d $out/w$w -o kt --output-format csv -- \
    kmertools_amd/bin/kmertools ctr -i /tmp/route_ab.fa -o /tmp/route_ab_$w -k 31 --devices $dev > $out/w$w.log 2>&1
  echo "== KT_ROUTE_WIDE=$w"; grep -E "route|scatter1w" $out/w$w/kt_kernel_stats.csv | awk -F, '{printf "   %-60s calls %s avg %.3f ms\n", substr($1,1,60), $2, $4/1e6}'
  tail -3 $out/w$w.log
done
sort /tmp/route_ab_1/kmers.counts | md5sum; sort /tmp/route_ab_0/kmers.counts | md5sum
