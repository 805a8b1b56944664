#!/bin/bash
# usage (GPU box): tools/profile_headline.sh <round tag, e.g. r6>  -> the default bench.py command (the driver's) under rocprofv3
# --kernel-trace --stats, its JSON line, and FETCH_SIZE / WRITE_SIZE of the oligo kernel from separate --pmc passes
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=${1:-r6}
out=gpurun_out/prof_${tag}_headline; rm -rf $out; mkdir -p $out
python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu > $out/bench_under_trace.json 2> $out/kt.err
{
  echo "# python3 bench.py --steps 20 --warmup 5   (the driver's command; JSON line of the plain run)"
  cat $out/bench.json
  echo "## the same command with --no-cpu under rocprofv3 --kernel-trace --stats: JSON line, then kernel stats"
  cat $out/bench_under_trace.json
  cat $out/kt_kernel_stats.csv
  echo "## the k=4 oligo kernel's launches of the traced run in order: the stats line above averages ALL of them - the 60 ramp"
  echo "## launches, the placement probes on up to eight output and eight input candidates, the launch-shape trials - so:"
  python3 - $out/kt_kernel_trace.csv <<'PY'
import csv, sys
d = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
           for r in csv.DictReader(open(sys.argv[1])) if "oligo_sb_kernel_dense<4" in r["Kernel_Name"])
ms = [x[1] for x in d]
timed = ms[-21:-1]   # --steps 20, then one launch of the untimed output check
print("oligo_sb_kernel_dense<4,...>: all %d launches: avg %.4f ms; the 20 launches of the timed region: avg %.4f ms, min %.4f, max %.4f" % (
    len(ms), sum(ms) / len(ms), sum(timed) / len(timed), min(timed), max(timed)))
PY
} > $out/summary.txt
cut -c1-300 $out/summary.txt | head -30
