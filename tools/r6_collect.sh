#!/bin/bash
# copies what tools/r6_profiles.sh left under gpurun_out/ into profiles/ (the committed names) and refreshes traffic.json
cd "$(dirname "$0")/.." || exit 1
for t in ctr_k31 ctr_k31_genome ctr_k15 ctr_k31_forced8 ctr_k31_forced8_genome comp_oligo_k4 comp_cgr_k7 headline; do
    d=gpurun_out/prof_r6_$t
    [ -f $d/summary.txt ] || continue
    cp $d/summary.txt profiles/r6_${t}_rocprof_summary.txt
    [ -f $d/kt_kernel_stats.csv ] && cp $d/kt_kernel_stats.csv profiles/r6_${t}_kernel_stats.csv
done
[ -f gpurun_out/prof_r6_headline/bench.json ] && cp gpurun_out/prof_r6_headline/bench.json profiles/r6_bench_headline.json
[ -s gpurun_out/r6_ctr_k31_pmc.txt ] && cp gpurun_out/r6_ctr_k31_pmc.txt profiles/r6_ctr_k31_pmc.txt
python3 tools/traffic_from_profiles.py r6 --write
