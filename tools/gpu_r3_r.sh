#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flag_fork or deferred or (step_matches_oracle and mixed)" 2>&1 | tail -4
AMD_SERIALIZE_KERNEL=3 timeout 300 python bench.py --steps 50 --warmup 5 --no_cpu_baseline --no_e2e --no_kernel_timing 2>&1 | tail -3 | cut -c1-300
timeout 300 bash tools/ab.sh 2 "" 2>&1 | tail -2
