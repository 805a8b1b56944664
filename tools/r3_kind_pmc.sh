#!/bin/bash
# usage (GPU box): tools/r3_kind_pmc.sh -> per copy of the bases: kernel ms (kernel trace) and counters, three --pmc passes
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r3_kind; rm -rf $out; mkdir -p $out
i=0
for pmc in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_REQUEST_sum" \
           "TCC_EA_RDREQ_sum TCC_EA_RDREQ_LEVEL_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_LEVEL_sum" \
           "TCC_EA_WRREQ_STALL_sum TCC_EA_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pmc -d $out/p$i -o r --output-format csv -- python3 tools/r3_kind_pmc.py > $out/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r3_kind/p?")):
    disp = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "oligo_sb" in r["Kernel_Name"]:
                disp[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    ctr = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "oligo_sb" in r["Kernel_Name"]:
                ctr[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(disp)
    print(d.split("/")[-1], len(ids), "dispatches")
    for g in range(0, len(ids), 12):
        grp = ids[g + 4:g + 12]
        if not grp: continue
        ms = sum(disp[i] for i in grp) / len(grp)
        names = sorted(ctr[grp[0]]) if grp[0] in ctr else []
        print("  copy %d  %.3f ms " % (g // 12, ms) + "  ".join("%s %.4g" % (nm.replace("_sum", ""), sum(ctr[i].get(nm, 0) for i in grp) / len(grp)) for nm in names))
PY
