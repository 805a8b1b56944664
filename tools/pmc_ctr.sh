#!/bin/bash
# usage (GPU box): tools/pmc_ctr.sh <tag> [bench args]: SQ / LDS counters of the ctr kernels (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/pmcctr_$tag; rm -rf $out; mkdir -p $out
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line -d $out/pass$i -o p --output-format csv -- python3 bench.py "$@" --no-cpu > $out/pass$i.log 2>&1
done <<'PASSES'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
PASSES
python3 tools/pmc_summary.py $out
