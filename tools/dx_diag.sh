#!/bin/bash
# diagnostic forms of the dX one-hot GEMM (library built with TCAR_HIPCC_FLAGS=-DTCAR_GEMM_DIAG): 0 = product, 1002 = no fills after
# stage 1, 1003 = fills + barriers only, 1006 = epilogue only, 1008 = K loop without the epilogue
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for t in 0 1002 1003 1006 1008; do
  echo -n "dx2 tile=$t: "; TCAR_BF16_TILE=$t python tools/gemm_bench.py dx2 1 50 2>&1 | grep -v Warning | tail -1
done; done
