#!/bin/bash
# kernel trace of the default bench (device held back while the host enqueues) -> idle time of the main stream in front of each kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_G -o tr -- python3 $R/bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing --no_by_T --stall_ms 150 "$@" > $R/gpurun_out/G_trace.log 2>&1 )
db=$(ls gpurun_out/prof_G/*/tr_results.db gpurun_out/prof_G/tr_results.db 2>/dev/null | head -1)
python tools/gap_stats.py $db | tee gpurun_out/gap_stats.txt
rm -rf gpurun_out/prof_G
