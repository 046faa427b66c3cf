#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TCAR_FLAG_FORK=1015 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flag or fork or same_step_twice or deferred or (step_matches_oracle and mixed)" 2>&1 | tail -2
timeout 1500 bash tools/ab.sh 5 "" "TCAR_FLAG_FORK=1015" 2>&1 | tee gpurun_out/r3r_ab22.txt
