#!/bin/bash
# timeline of one step under the settings given as arguments (env assignments), bf16x3-mixed
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for kv in "$@"; do export "$kv"; done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_h -o h -- python3 $OLDPWD/bench.py --scoring bf16x3-mixed --steps 200 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing > $OLDPWD/gpurun_out/prof_h.log 2>&1 )
db=$(ls gpurun_out/prof_h/*/h_results.db gpurun_out/prof_h/h_results.db 2>/dev/null | head -1)
python tools/kstats.py $db gpurun_out/prof_h_kstats.csv | head -12
python tools/timeline.py $db 100
