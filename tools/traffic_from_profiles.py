"""HBM traffic of one ctr step from a committed rocprofv3 summary (profiles/r5_<tag>_rocprof_summary.txt, the separate
FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh): per-dispatch means of the step's kernels x their launches per
step, summed.  Prints the fetch_kib / write_kib that profiles/traffic.json holds.  usage: traffic_from_profiles.py [tags...]"""
import pathlib, re, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
STEP_KERNELS = r"seg_index|scatter1|xcd_tails|part2|build_kernel|ext_scan|ext_patch|hist1|scan1"
for tag in sys.argv[1:] or ("ctr_k31", "ctr_k31_genome", "ctr_k15"):
    txt = (ROOT / "profiles" / ("r5_%s_rocprof_summary.txt" % tag)).read_text()
    rows = []
    for m in re.finditer(r"== (.+)\n((?:  .+\n)+)", txt[txt.index("## PMC"):]):
        if not re.search(STEP_KERNELS, m.group(1)):
            continue
        d = dict(re.findall(r"  (\w+)\s+mean ([0-9.e+]+)", m.group(2)))
        rows.append((m.group(1)[:44], float(d["FETCH_SIZE"]), float(d["WRITE_SIZE"]), int(re.search(r"\(n=(\d+)\)", m.group(2)).group(1))))
    steps = min(r[3] for r in rows if "build_kernel" in r[0])
    F = sum(r[1] * r[3] / steps for r in rows)
    W = sum(r[2] * r[3] / steps for r in rows)
    print("%s: %d steps profiled; fetch_kib %d write_kib %d -> %.1f GB per step (FETCH x 2 + WRITE)" % (tag, steps, F, W, (2 * F + W) * 1024 / 1e9))
    for r in rows:
        print("   %-46s reads %6.2f GB  writes %6.2f GB" % (r[0], 2 * r[1] * r[3] / steps * 1024 / 1e9, r[2] * r[3] / steps * 1024 / 1e9))
