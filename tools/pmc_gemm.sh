#!/bin/bash
# PMC passes (each counter set in its own run, kernel-trace only — MI355X_MICROARCH.md, HBM / rocprofv3) over
# tools/gemm_bench.py <mode> <nsplit>; raw output under gpurun_out/pmc_<mode>_n<nsplit>_<set>, packed by tools/pmc_pack.py
# (PMC_KERNEL=<substring>: the kernel to summarise when it is not gemm_bf16_kernel)
# usage: tools/pmc_gemm.sh <fwd|fwdce|fwdce2|dx|de|dx2|de2> [nsplit=3] [iters=5]   (fwdce2 = the step's form: one-hot time segment)
MODE=$1; NS=${2:-3}; IT=${3:-5}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  OUT=$ROOT/gpurun_out/pmc_${MODE}_n${NS}_${TAG}
  rm -rf $OUT
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT -- python3 $ROOT/tools/gemm_bench.py $MODE $NS $IT > $OUT.log 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT ${PMC_KERNEL:-gemm_bf16_kernel} 3 > $ROOT/gpurun_out/pmc_${MODE}_n${NS}_${TAG}.json 2>> $OUT.log
  find $OUT -name "*.csv" ! -name "*counter_collection.csv" -delete
done
python3 $ROOT/tools/pmc_pack.py $MODE $NS
