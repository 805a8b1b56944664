#!/bin/bash
# copies what tools/r5_profiles.sh left under gpurun_out/ into profiles/ (the committed names) and refreshes traffic.json
cd "$(dirname "$0")/.." || exit 1
for t in ctr_k31 ctr_k31_genome ctr_k15 ctr_k31_forced ctr_k31_forced_presplit headline; do
    d=gpurun_out/prof_r5_$t
    [ -f $d/summary.txt ] || continue
    cp $d/summary.txt profiles/r5_${t}_rocprof_summary.txt
    [ -f $d/kt_kernel_stats.csv ] && cp $d/kt_kernel_stats.csv profiles/r5_${t}_kernel_stats.csv
done
[ -f gpurun_out/prof_r5_headline/bench.json ] && cp gpurun_out/prof_r5_headline/bench.json profiles/r5_bench_headline.json
for f in r5_ctr_k31_pmc.txt r5_xcd_append_ubench.txt r5_store_overlap_ubench.txt; do
    [ -s gpurun_out/$f ] && cp gpurun_out/$f profiles/$f
done
python3 tools/traffic_from_profiles.py
