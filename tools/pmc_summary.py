#!/usr/bin/env python3
"""Averages rocprofv3 --pmc CSV output per kernel: tools/pmc_summary.py gpurun_out/pmc_<tag>"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        per[(name, row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    for (name, _), cs in per.items():
        for c, v in cs.items():
            agg[name][c].append(v)
for name, cs in agg.items():
    print("==", name)
    for c in sorted(cs):
        v = cs[c]
        print("  %-28s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
