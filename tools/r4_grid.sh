#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for g in 0 256 192 128 96 64 32; do
  echo "#### KT_P2_GRID=$g"
  KT_P2_GRID=$g tools/r4_abl.sh base "0" --steps 3 2>&1 | grep "part2_fast\|ms/step"
done
