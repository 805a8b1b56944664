#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_cli.py -x -q -m gpu -k "cov or out_of_core or many_batches" 2>&1 | tail -15
