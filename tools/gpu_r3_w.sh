#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "click_query or (step_matches_oracle and mixed) or deferred" 2>&1 | tail -3
timeout 600 bash tools/ab.sh 3 "" 2>&1 | tee gpurun_out/r3w_ab.txt
bash tools/gpu_r3_v.sh 2>&1 | grep -E "query_mlp|gemm_x3_kernel<0, 1, 64, 2>|attn_pool_fwd|poll_flag_kernel<false>|step span" | head -12
