#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flag_and_event or flag_forks_fall_back" 2>&1 | tail -1; done
timeout 300 bash tools/ab.sh 3 "" 2>&1 | tail -3
