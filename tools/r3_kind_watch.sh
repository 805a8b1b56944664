#!/bin/bash
# usage (GPU box): tools/r3_kind_watch.sh [seconds] -> one process launches oligo k=4 (32 workgroups per resident slot) back to
# back for a minute and a half, printing the mean per 100 launches with a wall-clock stamp; amd-smi clocks beside it.
# Do the two "kinds of process" alternate in time inside one process?
cd "$GRAFT_REPO_ROOT"
SECS=${1:-90}
KT_OLIGO_OVERSUB=32 python3 - $SECS <<'PY' &
import sys, time
sys.path.insert(0, ".")
import torch
from kmertools_amd import device
n, L = 10_000_000, 150
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
out = torch.empty((n, 136), dtype=torch.float64, device="cuda")
t_end = time.time() + float(sys.argv[1])
line = []
while time.time() < t_end:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(100): ctx.oligo(bases, offsets, n, 4, out)
    b.record(s); torch.cuda.synchronize()
    line.append("%.3f" % (a.elapsed_time(b) / 100))
    if len(line) == 8:
        print("t=%.1f oligo ms: %s" % (time.time() % 1000, " ".join(line)), flush=True); line = []
PY
pid=$!
sleep 4
amd-smi metric -c 2>/dev/null | head -60
while kill -0 $pid 2>/dev/null; do
  echo "t=$(python3 -c 'import time; print("%.1f" % (time.time() % 1000))') $(amd-smi metric -c -p 2>/dev/null | grep -E "SOCKET_POWER|CLK:|MIN_CLK|MAX_CLK" | tr -s ' ' | tr '\n' '|' | cut -c1-600)"
  sleep 2
done
wait $pid
