#!/bin/bash
# usage (GPU box): tools/r3_kind_watch2.sh [seconds] -> a long-running process (A) launches oligo k=4 at 32 workgroups per slot
# for the whole time, pausing 3 s every 20 s; in each pause a fresh short process (B) measures the same thing.  Is the "kind"
# a property of the process (A stays what it was while the Bs vary) or of the moment (A and the Bs flip together)?
cd "$GRAFT_REPO_ROOT"
SECS=${1:-240}
cat > /tmp/kw.py <<'PY'
import sys, time
sys.path.insert(0, ".")
import torch
from kmertools_amd import device
secs, tag = float(sys.argv[1]), sys.argv[2]
n, L = 10_000_000, 150
s = torch.cuda.current_stream()
ctx = device.Context(0, stream=s.cuda_stream)
bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads(1, n, L, bases, offsets)
out = torch.empty((n, 136), dtype=torch.float64, device="cuda")
def burst(k):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(k): ctx.oligo(bases, offsets, n, 4, out)
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
t0 = time.time()
if tag == "B":
    burst(100)
    print("   B t=%.0f: %.3f %.3f ms" % (time.time() % 10000, burst(200), burst(200)), flush=True)
    sys.exit(0)
while time.time() - t0 < secs:
    t1 = time.time(); r = []
    while time.time() - t1 < 17: r.append(burst(400))
    print("A t=%.0f: min %.3f max %.3f last %.3f ms" % (time.time() % 10000, min(r), max(r), r[-1]), flush=True)
    time.sleep(3.2)
PY
export KT_OLIGO_OVERSUB=32
python3 /tmp/kw.py $SECS A &
pid=$!
sleep 12
while kill -0 $pid 2>/dev/null; do
  sleep 6.5
  python3 /tmp/kw.py 0 B 2>/dev/null
  sleep 10
done
wait $pid
