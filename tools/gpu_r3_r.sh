#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flag or same_step_twice or deferred or (step_matches_oracle and mixed) or gemm_grouped or gemm_epilogues" 2>&1 | tail -2
timeout 1500 bash tools/ab.sh 5 "" "TCAR_DBG_FLUSH=1" 2>&1 | tee gpurun_out/r3r_ab16.txt
