#!/bin/bash
# click-query chain on the third stream (flag fork + flag join): parity, A/B, timeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export TCAR_FLAG_FORK=240
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "step_matches_oracle or deferred or same_step_twice or full_size or python_sequenced or gather_forward or click_query" 2>&1 | tail -5
unset TCAR_FLAG_FORK
timeout 900 bash tools/ab.sh 3 "TCAR_FLAG_FORK=48" "TCAR_FLAG_FORK=240" 2>&1 | tee gpurun_out/r3t_ab.txt
( cd /tmp && export TMPDIR=/tmp && export TCAR_FLAG_FORK=240 && timeout 200 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_r3t -o r3t -- python3 $OLDPWD/bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing > $OLDPWD/gpurun_out/prof_r3t.log 2>&1 )
db=$(find gpurun_out/prof_r3t -name "*.db" | head -1)
[ -n "$db" ] && timeout 100 python tools/timeline.py $db 60 < /dev/null > gpurun_out/r3t_timeline.txt
head -24 gpurun_out/r3t_timeline.txt; tail -1 gpurun_out/r3t_timeline.txt
rm -rf gpurun_out/prof_r3t
