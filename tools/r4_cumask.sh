#!/bin/bash
# GPU box: ctr k=31 kernels with the process restricted to a subset of the CUs (is a pass per-CU bound or system bound?)
cd "$GRAFT_REPO_ROOT" || exit 1
F=ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff
H=ffffffffffffffffffffffffffffffff
A=5555555555555555555555555555555555555555555555555555555555555555
for m in "" 0x$F 0x$H 0x$A; do
  echo "#### ROC_GLOBAL_CU_MASK=$m"
  if [ -z "$m" ]; then tools/r4_abl.sh base "0" --steps 3; else ROC_GLOBAL_CU_MASK=$m tools/r4_abl.sh base "0" --steps 3; fi
done
