#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
timeout 100 python tools/gemm_bench.py fwdce2 3 50 2>&1 | tail -1
TCAR_DBG_NOPLANE=1 timeout 100 python tools/gemm_bench.py fwdce2 3 50 2>&1 | tail -1
done
