#!/bin/bash
# flag forks: parity tests, A/B against event forks, timeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deferred or same_step_twice" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_e2e.py tests/test_gpu_sampler.py -x -q -m gpu 2>&1 | tail -8
timeout 600 bash tools/ab.sh 3 "" "TCAR_FLAG_FORK=0" 2>&1 | tee gpurun_out/r3p_ab.txt
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_r3p -o r3p -- python3 $OLDPWD/bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing > $OLDPWD/gpurun_out/prof_r3p.log 2>&1 )
db=$(find gpurun_out/prof_r3p -name "*.db" | head -1)
[ -n "$db" ] && timeout 100 python tools/timeline.py $db 60 < /dev/null > gpurun_out/r3p_timeline.txt
cat gpurun_out/r3p_timeline.txt
tail -3 gpurun_out/prof_r3p.log
rm -rf gpurun_out/prof_r3p
