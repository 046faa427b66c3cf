#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm_bf16 or softmax_epilogue" 2>&1 | tail -2
for m in fwdce2 fwdce; do timeout 120 python tools/gemm_bench.py $m 3 50 2>&1 | tail -1; done
timeout 120 python tools/gemm_bench.py fwd 3 50 2>&1 | tail -1
timeout 600 bash tools/ab.sh 3 "" 2>&1 | tail -3
