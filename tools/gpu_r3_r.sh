#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TCAR_FLAG_FORK=250 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "step_matches_oracle or deferred or same_step_twice or full_size or negative_modes or bit_identical" 2>&1 | tail -3
timeout 1500 bash tools/ab.sh 5 "" "TCAR_FLAG_FORK=250" 2>&1 | tee gpurun_out/r3r_ab12.txt
