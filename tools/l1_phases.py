"""Phase timers of level 1 and the range build (a library compiled with -DKT_ABLATION=1: tools/build_variant_tu.sh abl kt_bulk
-DKT_ABLATION=1; KT_LIB names it): cycles of thread 0 of every workgroup between the KT_PH marks of scatter1y_kernel /
build_kernel, summed over a counting step, as shares.  Timing builds only - the shipped library carries no timers.
usage (GPU box): KT_LIB=.../variants/libabl.so python3 tools/l1_phases.py <k> [reads, default 12000000]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from kmertools_amd import _lib, device  # noqa: E402

L1 = ["open unit", "count (+2 stores)", "barrier", "scan + cursor atomic (+1 store)", "barrier", "placement (+2 stores)",
      "pads, delta, take prefetch + barrier", "stop flag"]
BUILD = ["clear", "barrier", "insert", "barrier", "pack", "barrier", "copy-out", "barrier"]


def main():
    k = int(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12_000_000
    Lr = 150
    ctx = device.Context(0, torch.cuda.current_stream().cuda_stream)
    bases = torch.empty(n * Lr, dtype=torch.uint8, device="cuda")
    offs = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_reads(20240531, n, Lr, bases, offs)
    distinct = min(n * (Lr - k + 1), (4 ** k + 2 ** k) // 2)
    ctr = device.Counter(ctx, k, int(1.9 * distinct))
    xk = torch.empty(distinct, dtype=torch.int64, device="cuda")   # (like bench.py: the build writes the export arrays)
    xc = torch.empty(distinct, dtype=torch.int32, device="cuda")
    ctr.export_target(xk, xc, distinct)
    lib = _lib.lib()
    out = (C.c_ulonglong * 16)()
    for i in range(3):
        ctr.clear()
        ctr.add_reads(bases, offs, n)
        torch.cuda.synchronize()
        if i == 1:
            lib.kt_dbg_phases(out)  # (reads and zeroes: the last step alone is reported)
    lib.kt_dbg_phases(out)
    for name, labels, lo in (("level 1", L1, 8), ("build", BUILD, 0)):
        tot = sum(out[lo:lo + 8]) or 1
        print("%s (k=%d, %d reads): %s" % (name, k, n, "  ".join("%s %.1f%%" % (labels[i], 100.0 * out[lo + i] / tot) for i in range(8))))


if __name__ == "__main__":
    main()
