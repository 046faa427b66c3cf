#!/bin/bash
# timeline of a T = 40 step (B = 512: 20,480 rows): the last steps of a traced bench run are its ms_per_step_by_T loop at --by_T 40
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_T40 -o tr -- python3 $R/bench.py --steps 30 --warmup 10 --no_cpu_baseline --no_e2e --by_T ${1:-40} > $R/gpurun_out/T40_trace.log 2>&1 )
db=$(ls gpurun_out/prof_T40/*/tr_results.db gpurun_out/prof_T40/tr_results.db 2>/dev/null | head -1)
python tools/timeline.py $db -12 clip_adam_early > gpurun_out/T40_timeline.txt
# per-kernel statistics of the LAST 60 steps of the trace = the by_T loop's own steps (its 15 + 2 x 40 steps end the run)
python tools/kstats.py $db gpurun_out/T40_kernel_stats.csv ${2:-60} > gpurun_out/T40_kernel_stats.txt
cat gpurun_out/T40_timeline.txt
rm -rf gpurun_out/prof_T40
