#!/bin/bash
# LDS counters of one scoring GEMM alone (own --pmc pass, kernel-trace only): bank-conflict cycles against all LDS-array cycles.
# usage: tools/pmc_lds.sh <fwdce2|dx2|de2> [nsplit]
MODE=$1; NS=${2:-1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_lds_${MODE}
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT -- python3 $ROOT/tools/gemm_bench.py $MODE $NS 5 > $OUT.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT gemm_bf16_kernel 3
