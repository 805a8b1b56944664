#!/usr/bin/env python3
"""Build-time check of the kernels that issue loads from inline assembly (kt_bulk.hip: buf_load8_async / buf_wait /
buf_take; kt_segment.hpp: prefetch_issue / prefetch_take): between an asm load into a register and the asm s_waitcnt behind it, no instruction of the compiler's
own may name that register - a copy or a spill made there would move a value that has not arrived.

usage: check_inflight.py <device assembly .s> <kernel name regex> [<regex of the kernels whose counted waits are checked>]

The walk is in layout order, which is the order of execution for these loops (the load block lies in front of the
wait block in the chunk loop's body; code the compiler placed out of line is not followed).  Exit status 1 and a
listing when something is found.

A wait `s_waitcnt vmcnt(N)` with N > 0 is only sound when at least N vector-memory operations are issued between the asm
loads and the wait on every path (the loads have landed once at most N YOUNGER operations are outstanding).  The walk
counts the vector-memory instructions it passes between the last asm load and the wait and fails when there are fewer than
N - which catches a store that was removed or moved, not one that was put under a condition (the count is of the layout,
not of a path): the kernels issue those stores unconditionally by construction (a group with nothing to write goes to a
dump line), and the GPU suite is what checks that.  Only for the kernels the third argument names: where the loads sit at
the bottom of a loop and the wait at its top (the level-2 kernels) the layout between them is not what runs between them."""
import re
import sys


def regs_of(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bv(\d+)\b", text))
    return out


def dead_high_addend(lines, i, code, inflight):
    """v_mad_u64_u32 D, sdst, a, b, v[lo:hi] with only `hi` in flight: the compiler's way to write lo32(a * b) + x - the
    addend pair is {x, whatever lies beside it}, and that half reaches only D's high half.  Accepted when D's high half is
    written again before anything reads it."""
    m = re.match(r"v_mad_u64_u32\s+v\[(\d+):(\d+)\],\s*[^,]+,\s*[^,]+,\s*[^,]+,\s*v\[(\d+):(\d+)\]\s*$", code.strip())
    if not m:
        return False
    dhi, alo, ahi = int(m.group(2)), int(m.group(3)), int(m.group(4))
    others = regs_of(code.rsplit(",", 1)[0])  # everything but the addend pair
    if ahi not in inflight or alo in inflight or (others & inflight):
        return False
    for nxt in lines[i + 1:i + 200]:
        t = nxt.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        c = t.split(";")[0]
        if re.match(r"\S+:", c):      # a label: give up looking, refuse
            return False
        ops = c.split(None, 1)
        if len(ops) < 2:
            continue
        first, _, rest = ops[1].partition(",")
        if dhi in regs_of(rest):      # read before being written again
            return False
        if dhi in regs_of(first):     # (destination operand comes first) written again: dead
            return True
    return False


def check(path, pattern, count_pattern=None):
    bad = []
    fn = None
    in_asm = False
    inflight = set()
    pending_clear = False
    n_loads = n_waits = 0
    vmem_since = 0      # vector-memory instructions passed since the last asm load into a register that is still in flight
    lines = open(path).read().split("\n")
    for ln, line in enumerate(lines, 1):
        s = line.strip()
        m = re.match(r"^(\S+):\s*(;.*)?$", line)
        if m and not m.group(1).startswith(".L"):
            fn = m.group(1) if re.search(pattern, m.group(1)) else None
            inflight.clear()
            in_asm = False
            continue
        if fn is None:
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            if pending_clear:
                inflight.clear()
                pending_clear = False
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        code = s.split(";")[0]
        if in_asm:
            m = re.match(r"(?:buffer|global)_load_dword(x\d)?\s+(v\[\d+:\d+\]|v\d+)\s*,", code)
            if m:
                inflight |= regs_of(m.group(2))
                n_loads += 1
                vmem_since = 0
            elif re.match(r"s_waitcnt\s+vmcnt", code):
                mw = re.match(r"s_waitcnt\s+vmcnt\((\d+)\)", code)
                if mw and inflight and count_pattern and re.search(count_pattern, fn) and int(mw.group(1)) > vmem_since:
                    bad.append((fn, ln, "%s behind only %d vector-memory instruction(s)" % (code.strip(), vmem_since), sorted(inflight)[:4]))
                pending_clear = True
                n_waits += 1
            elif re.match(r"(?:buffer|global|flat|scratch)_(?:load|store|atomic)", code):
                vmem_since += 1
            continue
        if re.match(r"s_waitcnt\s+vmcnt\(0\)", code):  # (one of the compiler's own: everything has landed)
            inflight.clear()
            continue
        if re.match(r"\s*(?:buffer|global|flat|scratch)_(?:load|store|atomic)", code):
            vmem_since += 1
        hit = regs_of(code) & inflight
        if hit and dead_high_addend(lines, ln - 1, code, inflight):
            hit = set()
        if hit:
            bad.append((fn, ln, code.strip(), sorted(hit)))
    return bad, n_loads, n_waits


if __name__ == "__main__":
    bad, n_loads, n_waits = check(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    if n_loads == 0 or n_waits == 0:
        print("check_inflight: no asm loads / waits found in kernels matching %r - the check is not looking at anything" % sys.argv[2])
        sys.exit(1)
    for fn, ln, code, hit in bad:
        print("register in flight touched: %s line %d: %s   (v%s)" % (fn, ln, code, ", v".join(map(str, hit))))
    if bad:
        sys.exit(1)
    print("check_inflight: %d asm loads, %d asm waits, no in-flight register touched" % (n_loads, n_waits))
