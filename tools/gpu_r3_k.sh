#!/bin/bash
# round 3, session K: fewer events on the main stream (A/B), correctness of the reordered prologue
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_e2e.py -m gpu -q --tb=short -x -k "deferred or golden or globo_full_size or bit_identical or same_step_twice or without_negatives or (step_matches_oracle and mixed)" 2>&1 | grep -v "^$" | tail -5
bash tools/ab.sh 3 "" "TCAR_FEW_EVENTS=0" 2>&1 | tee gpurun_out/r3k_ab.txt
