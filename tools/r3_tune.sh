#!/bin/bash
# usage (GPU box): tools/r3_tune.sh -> the launch-shape measurement of oligo k=4: its test, then the headline workload in
# several processes with the measurement on, with it off (96 per slot) and with 32 fixed
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "launch_shape or cfg2_full" 2>&1 | tail -3
for i in 1 2 3 4; do
  python bench.py --workload comp_oligo_k4 --steps 20 --warmup 5 --no-cpu > gpurun_out/r3_tune_on$i.json 2>/dev/null
  KT_OLIGO_TUNE=0 python bench.py --workload comp_oligo_k4 --steps 20 --warmup 5 --no-cpu > gpurun_out/r3_tune_off$i.json 2>/dev/null
  KT_OLIGO_OVERSUB=32 python bench.py --workload comp_oligo_k4 --steps 20 --warmup 5 --no-cpu > gpurun_out/r3_tune_32_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3_tune_*.json")):
    for ln in open(f):
        if ln.startswith("{"):
            r = json.loads(ln)
            print(f.split("/")[-1], r["ms_per_step"], r["roofline"]["frac"], r.get("oligo_launch"), r.get("environment", {}).get("KT_OLIGO_OVERSUB"))
PY
