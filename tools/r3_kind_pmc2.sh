#!/bin/bash
# usage (GPU box): tools/r3_kind_pmc2.sh -> oligo k=4 with 32 workgroups per resident slot (where the two kinds of process
# differ most), several processes per counter set: which counters separate the fast processes from the slow ones?
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r3_kind2; rm -rf $out; mkdir -p $out
export KT_OLIGO_OVERSUB=32
i=0
for pmc in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_LFIFO_FULL GRBM_UTCL2_BUSY" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_WRITE_sum TCC_WRITEBACK_sum"; do
  i=$((i+1))
  for p in 1 2 3 4 5 6 7 8; do
    rocprofv3 --kernel-trace --pmc $pmc -d $out/c${i}_$p -o r --output-format csv -- python3 bench.py --workload comp_oligo_k4 --steps 5 --warmup 2 --no-cpu > $out/c${i}_$p.json 2> $out/c${i}_$p.err
  done
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r3_kind2/c?_?")):
    acc = collections.defaultdict(list); dur = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "oligo_sb" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "oligo_sb" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    dur = dur[len(dur) // 2:]
    print(d.split("/")[-1], "ms %.3f " % (sum(dur) / max(1, len(dur))) + "  ".join("%s %.4g" % (k.replace("_sum", ""), sum(v[len(v)//2:]) / max(1, len(v[len(v)//2:]))) for k, v in sorted(acc.items())))
PY
