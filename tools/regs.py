#!/usr/bin/env python3
"""tools/regs.py <file.hip> [name filter] [-D...]: compiles the file device-only for gfx950 (here, no GPU needed) and
prints VGPR / AGPR / SGPR / spill / LDS / scratch per kernel - the occupancy inputs - from the assembly's metadata."""
import re
import subprocess
import sys

src = sys.argv[1]
filt = [a for a in sys.argv[2:] if not a.startswith("-")]
defs = [a for a in sys.argv[2:] if a.startswith("-")]
out = "/tmp/regs_%s.s" % re.sub(r"\W", "_", src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                       src, "-o", out] + defs, stderr=subprocess.DEVNULL)
t = open(out).read()
md = t[t.rfind("amdhsa.kernels:"):]
for blk in re.split(r"\n  - ", md)[1:]:
    def f(key):
        m = re.search(r"\." + key + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = subprocess.run(["c++filt", f("name")], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0]
    if filt and not any(x in name for x in filt):
        continue
    print("%-70s vgpr %3s agpr %3s sgpr %3s spill %s/%s lds %6s scratch %s" % (
        name[:70], f("vgpr_count"), f("agpr_count"), f("sgpr_count"), f("vgpr_spill_count"), f("sgpr_spill_count"),
        f("group_segment_fixed_size"), f("private_segment_fixed_size")))
