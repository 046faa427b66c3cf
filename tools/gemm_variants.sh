#!/bin/bash
# Interleaved timing of kernel VARIANTS of the logits GEMM (TCAR_BF16_TILE codes), each with an output checksum.
#   tools/gemm_variants.sh "<code> <code> ..." [rounds=3] [mode=fwdce2]
cd ${GRAFT_REPO_ROOT:-/root/repo}
CODES=$1; R=${2:-3}; MODE=${3:-fwdce2}
for r in $(seq $R); do for c in $CODES; do
  GB_SUM=1 TCAR_BF16_TILE=$c python tools/gemm_bench.py $MODE 3 50 2>&1 | grep -v Warning | tail -1
done; done
