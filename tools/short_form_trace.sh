#!/bin/bash
# kernel trace of the driver's short form (20 steps, 5 warm-up, headline loop only) -> per-step periods (tools/step_periods.py)
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_sf -o tr -- python3 $R/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_e2e --no_kernel_timing --no_by_T > $R/gpurun_out/sf_trace.log 2>&1 )
db=$(ls gpurun_out/prof_sf/*/tr_results.db gpurun_out/prof_sf/tr_results.db 2>/dev/null | head -1)
python tools/step_periods.py $db > gpurun_out/sf_periods.txt
cat gpurun_out/sf_periods.txt; tail -1 gpurun_out/sf_trace.log | cut -c1-200
rm -rf gpurun_out/prof_sf
