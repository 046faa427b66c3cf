#!/bin/bash
# The logits GEMM's two MFMA forms (TCAR_LOGITS_MFMA16 = 0 | 1), interleaved: parity test of the softmax epilogue in the 16 x 16 x 32 form,
# then the kernel alone (tools/gemm_bench.py fwdce2: the step's form with the one-hot segment) with output checksums.
cd ${GRAFT_REPO_ROOT:-/root/repo}
TCAR_LOGITS_MFMA16=1 timeout 900 python -m pytest tests -m gpu -x -q --tb=short -k "logits_gemm_softmax_epilogue or globo_full_size" 2>&1 | tail -6
for r in 1 2 3; do for v in 0 1; do
  echo -n "TCAR_LOGITS_MFMA16=$v  "; GB_SUM=1 TCAR_LOGITS_MFMA16=$v python tools/gemm_bench.py fwdce2 3 50 2>&1 | grep -v Warning | tail -1
done; done
