#!/bin/bash
# one kernel trace of the default bench (device held back while the host enqueues), timelines of a T = 7, a T = 4 and a T = 1 step
# (the schedule is the 48-batch cycle of bench.py: the k-th gather of the run is batch k % 48; batch 10 has T = 7, 9 has T = 4, 11 T = 1)
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_T -o tr -- python3 $R/bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing --no_by_T --stall_ms 150 > $R/gpurun_out/T_trace.log 2>&1 )
db=$(ls gpurun_out/prof_T/*/tr_results.db gpurun_out/prof_T/tr_results.db 2>/dev/null | head -1)
for k in 106 105 107; do echo "== step $k"; python tools/timeline.py $db $k; done > gpurun_out/T_timelines.txt
rm -rf gpurun_out/prof_T
