#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in fwdce2 dx de; do timeout 120 python tools/gemm_bench.py $m $([ $m = fwdce2 ] && echo 3 || echo 1) 50 2>&1 | tail -1; done
timeout 600 bash tools/ab.sh 3 "" 2>&1 | tail -3
