#!/bin/bash
# round 3, session J: which other HBM-streaming passes should bypass the caches (non-temporal): A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TCAR_NT=61 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "deferred or golden or globo_full_size" 2>&1 | grep -v "^$" | tail -4
bash tools/ab.sh 3 "TCAR_NT=1" "TCAR_NT=5" "TCAR_NT=9" "TCAR_NT=17" "TCAR_NT=33" "TCAR_NT=61" 2>&1 | tee gpurun_out/r3j_ab.txt
