#!/bin/bash
# usage (GPU box): tools/clock_watch.sh  -> runs ~6 s of back-to-back oligo k=4 launches and samples rocm-smi clocks / power
# beside them (is the box's sustained shader clock the reason for the 2.0 / 2.3 ms split between boxes?)
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY' &
import sys, pathlib, time
sys.path.insert(0, ".")
import torch
from kmertools_amd import device
n, L = 10_000_000, 150
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
out = torch.empty((n, 136), dtype=torch.float64, device="cuda")
for rep in range(6):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(400): ctx.oligo(bases, offsets, n, 4, out)
    b.record(s); torch.cuda.synchronize()
    print("oligo: %.3f ms per launch (400 launches)" % (a.elapsed_time(b) / 400), flush=True)
x = torch.empty(1 << 30, dtype=torch.int64, device="cuda")
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(300): x.fill_(1)
    b.record(s); torch.cuda.synchronize()
    print("fill: %.0f GB/s (300 x 8 GiB)" % (300 * 8 * (1 << 30) / (a.elapsed_time(b) * 1e-3) / 1e9), flush=True)
PY
pid=$!
sleep 3
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  amd-smi metric -c -p 2>/dev/null | grep -E "SOCKET_POWER|THROTTLE|^ +CLK:" | head -10 | tr -s ' ' | tr '\n' '|'; echo
  sleep 0.7
done
wait $pid
