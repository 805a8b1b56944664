#!/bin/bash
# usage: tools/pmc.sh <tag> <what> [launches]   (run on the GPU box, from the repo root)
# Collects rocprofv3 PMC counters for one hot-path kernel in separate passes (no tracing
# domains combined with --pmc) into gpurun_out/pmc_<tag>/passN/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; what=${2:-oligo}; reps=${3:-3}
out=gpurun_out/pmc_$tag; mkdir -p $out
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line -d $out/pass$i -o p --output-format csv -- python3 tools/prof_oligo.py $what $reps > $out/pass$i.log 2>&1
done <<'PASSES'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_LDS_ADDR_CONFLICT
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL
PASSES
python3 tools/pmc_summary.py $out
