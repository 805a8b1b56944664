#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for k in 31 15; do
for g in 0 192 128; do
  echo "#### k=$k KT_P2_GRID=$g"
  if [ $k = 31 ]; then a="--k 31 --reads 25000000"; else a="--k 15 --reads 50000000"; fi
  KT_P2_GRID=$g tools/r4_abl.sh base "0" --steps 2 $a 2>&1 | grep "part2_swwc\|ms/step\|scatter1w"
done; done
