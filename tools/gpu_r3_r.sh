#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fork_state_does_not_leak or flag_and_event or flag_forks_fall_back" 2>&1 | tail -1; done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
