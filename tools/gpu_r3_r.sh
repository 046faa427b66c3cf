#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "(step_matches_oracle and mixed) or deferred or split_adam" 2>&1 | tail -3
TCAR_FLAG_FORK=251 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "(step_matches_oracle and mixed) or deferred or split_adam or (full_size and mixed)" 2>&1 | tail -3
timeout 1500 bash tools/ab.sh 4 "" "TCAR_FLAG_FORK=243" "TCAR_FLAG_FORK=250" "TCAR_FLAG_FORK=251" 2>&1 | tee gpurun_out/r3r_ab8.txt
