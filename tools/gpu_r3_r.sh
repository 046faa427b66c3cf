#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
timeout 300 bash tools/ab.sh 2 "" 2>&1 | tail -2
