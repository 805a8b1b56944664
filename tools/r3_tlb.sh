#!/bin/bash
# usage (GPU box): tools/r3_tlb.sh -> address-translation counters of the oligo k=4 kernel with 32 and with 96 workgroups
# per resident slot (separate --pmc passes, kernel trace beside them for the durations)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r3_tlb; rm -rf $out; mkdir -p $out
for s in 32 96; do
  export KT_OLIGO_OVERSUB=$s
  i=0
  for pmc in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
             "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
             "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_UTCL1_LFIFO_FULL TCP_UTCL1_STALL_INFLIGHT_MAX_sum"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $pmc -d $out/s${s}_p$i -o r --output-format csv -- python3 bench.py --workload comp_oligo_k4 --steps 5 --warmup 2 --no-cpu > $out/s${s}_p$i.json 2> $out/s${s}_p$i.err
  done
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r3_tlb/s*_p?")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "oligo_sb" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "oligo_sb" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    dur = dur[len(dur) // 2:]
    print(d.split("/")[-1], "ms %.3f" % (sum(dur) / max(1, len(dur))), {k: "%.4g" % (sum(v[len(v)//2:]) / max(1, len(v[len(v)//2:]))) for k, v in acc.items()})
PY
