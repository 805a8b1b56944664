"""HBM traffic per step / launch from the committed rocprofv3 summaries (profiles/<round>_<tag>_rocprof_summary.txt: the
separate FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh): per-dispatch means of a workload's kernels x their
launches per step, summed.  With --write the figures go into profiles/traffic.json (what bench.py's roofline.traffic reads),
each with the summary it came from and the commit the tree was at when the profile was collected.
usage: traffic_from_profiles.py <round, e.g. r6> [--write]"""
import json
import pathlib
import re
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
CTR_KERNELS = r"seg_index|pack_segments|route_kernel|scatter1|xcd_tails|part2|build_kernel|ext_scan|ext_patch|hist1|scan1"
# traffic.json key -> (summary tag, kernels of the workload's step, kernel whose launch count = steps (None: per launch of it))
WORKLOADS = {
    "ctr_k31": ("ctr_k31", CTR_KERNELS, "build_kernel"),
    "ctr_k31_genome": ("ctr_k31_genome", CTR_KERNELS, "build_kernel"),
    "ctr_k15": ("ctr_k15", CTR_KERNELS, "build_kernel"),
    "ctr_k31_forced8": ("ctr_k31_forced8", CTR_KERNELS, "build_kernel"),
    "comp_oligo_k4": ("comp_oligo_k4", r"oligo_sb_kernel_dense<4", None),
    "comp_cgr_k7": ("comp_cgr_k7", r"oligo_pw_kernel<7", None),
}


def rows_of(txt, pattern):
    rows = []
    for m in re.finditer(r"== (.+)\n((?:  .+\n)+)", txt[txt.index("## PMC"):]):
        if not re.search(pattern, m.group(1)):
            continue
        d = dict(re.findall(r"  (\w+)\s+mean ([0-9.e+]+)", m.group(2)))
        if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
            continue
        rows.append((m.group(1)[:60], float(d["FETCH_SIZE"]), float(d["WRITE_SIZE"]), int(re.search(r"\(n=(\d+)\)", m.group(2)).group(1))))
    return rows


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r6"
    write = "--write" in sys.argv
    tj_path = ROOT / "profiles" / "traffic.json"
    tj = json.loads(tj_path.read_text())
    try:
        head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except OSError:
        head = ""
    for key, (tag, pattern, per_step) in WORKLOADS.items():
        path = ROOT / "profiles" / ("%s_%s_rocprof_summary.txt" % (rnd, tag))
        if not path.exists():
            continue
        rows = rows_of(path.read_text(), pattern)
        if not rows:
            print("%s: no counters in %s" % (key, path.name))
            continue
        if per_step:
            steps = min(r[3] for r in rows if per_step in r[0])
            F = sum(r[1] * r[3] / steps for r in rows)
            W = sum(r[2] * r[3] / steps for r in rows)
        else:   # one kernel, per launch (its dispatches' mean)
            steps = rows[0][3]
            F, W = rows[0][1], rows[0][2]
        print("%s: %d step(s) / launches profiled; fetch_kib %d write_kib %d -> %.1f GB (FETCH x 2 + WRITE)" % (
            key, steps, F, W, (2 * F + W) * 1024 / 1e9))
        for r in rows:
            k = r[3] / steps if per_step else 1
            print("   %-62s reads %6.2f GB  writes %6.2f GB" % (r[0], 2 * r[1] * k * 1024 / 1e9, r[2] * k * 1024 / 1e9))
        if write:
            old = tj.get(key, {})
            tj[key] = {"fetch_kib": int(F), "write_kib": int(W), "fetch_correction": old.get("fetch_correction", 2),
                       "source": "profiles/" + path.name, "profile_head": head,
                       "per_kernel_gb": {r[0].split("<")[0].split("(")[0]: [round(2 * r[1] * (r[3] / steps if per_step else 1) * 1024 / 1e9, 2),
                                                                             round(r[2] * (r[3] / steps if per_step else 1) * 1024 / 1e9, 2)] for r in rows}}
    if write:
        tj_path.write_text(json.dumps(tj, indent=1) + "\n")
        print("wrote", tj_path)


if __name__ == "__main__":
    main()
