"""prints value / ms per step / roofline fraction of every bench line given (headline lines: every object)"""
import json
import sys

for path in sys.argv[1:]:
    try:
        j = json.loads([ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1])
    except Exception as e:  # noqa: BLE001
        print(path, "unreadable:", e)
        continue
    rows = [("top", j)] + [(k, v) for k, v in j.items() if isinstance(v, dict) and "roofline" in v]
    for name, o in rows:
        print("%-40s %-14s %9.2f Gbases/s %9.3f ms (median %s min %s) frac %.4f" % (
            path.split("/")[-1], name, o["value"], o["ms_per_step"], o.get("ms_median"), o.get("ms_min"), o["roofline"]["frac"]))
        g = o.get("genome_sampled")
        if g:
            print("%-40s %-14s %9.2f Gbases/s %9.3f ms frac %.4f" % ("", name + " genome", g["value"], g["ms_per_step"], g["roofline"]["frac"]))
