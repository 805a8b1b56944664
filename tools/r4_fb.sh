#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in base fb1 fb2 fb4; do
  lib=$PWD/kmertools_amd/variants/lib$v.so; [ $v = base ] && lib=$PWD/kmertools_amd/libkmertools_hip.so
  echo "== $v"; KT_LIB=$lib python tools/r4_dbg1.py 2>&1 | grep "'KT_S1_COMB': '1'}" | cut -c20-
  KT_LIB=$lib timeout 300 python tools/r4_ctr_run.py --steps 2 2>&1 | tail -1
done
