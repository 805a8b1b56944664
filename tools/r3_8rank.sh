#!/bin/bash
# eight ranks sharing one GPU (gloo host transport): the 8-GPU geometry (ownership intervals, 4 slices x 8 senders of sources,
# the pre-split pass forced), checked by bench.py's own output check (the counts of all ranks sum to reads x (L - k + 1))
cd "$GRAFT_REPO_ROOT" || exit 1
export KT_BENCH_SHARE_GPU=1
for mb in 11 7; do
  echo "### 8 ranks, KT_BULK_MAX_B2=$mb"
  KT_BULK_MAX_B2=$mb KT_BULK_VERBOSE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 \
     bench.py --gpus 8 --workload ctr_k31 --reads 1000000 --steps 2 --warmup 1 --no-cpu > gpurun_out/r3_8rank_$mb.json 2> gpurun_out/r3_8rank_$mb.err
  echo "rc=$?"; grep -c "presplit" gpurun_out/r3_8rank_$mb.err; grep -E "Error|rror|assert" gpurun_out/r3_8rank_$mb.err | head -3
  python - <<PY
import json
ln=[l for l in open("gpurun_out/r3_8rank_$mb.json") if l.startswith("{")]
if ln:
    j=json.loads(ln[-1]); print({k:j[k] for k in ("n_gpus","value","ms_per_step","output_check")}, j["config"]["parallelism"][-60:])
PY
done
echo "### 3 ranks k=15"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29542 \
     bench.py --gpus 3 --workload ctr_k15 --reads 1000000 --steps 2 --warmup 1 --no-cpu > gpurun_out/r3_3rank.json 2> gpurun_out/r3_3rank.err; echo "rc=$?"
python - <<PY
import json
ln=[l for l in open("gpurun_out/r3_3rank.json") if l.startswith("{")]
if ln:
    j=json.loads(ln[-1]); print({k:j[k] for k in ("n_gpus","value","ms_per_step","output_check")})
PY
