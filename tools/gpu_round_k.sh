#!/bin/bash
# timeline of the catalog-sharded step on one rank (no collectives)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export TCAR_FORCE_DP=1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_k -o k -- python3 $OLDPWD/bench.py --dp_mode ${1:-sharded} --steps 200 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing > $OLDPWD/gpurun_out/prof_k.log 2>&1 )
db=$(ls gpurun_out/prof_k/*/k_results.db gpurun_out/prof_k/k_results.db 2>/dev/null | head -1)
python tools/kstats.py $db gpurun_out/prof_k_kstats.csv | head -14
python tools/timeline.py $db 100
