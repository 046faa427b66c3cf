#!/bin/bash
# timeline of the step with the one-hot time segment + A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "time_onehot or (full_size and mixed) or deferred" 2>&1 | tail -5
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_r3m -o r3m -- python3 $OLDPWD/bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing > $OLDPWD/gpurun_out/prof_r3m.log 2>&1 )
db=$(ls gpurun_out/prof_r3m/*/r3m_results.db gpurun_out/prof_r3m/r3m_results.db 2>/dev/null | head -1)
[ -n "$db" ] && timeout 100 python tools/timeline.py $db 60 < /dev/null > gpurun_out/r3m_timeline.txt
cat gpurun_out/r3m_timeline.txt
rm -rf gpurun_out/prof_r3m
timeout 600 bash tools/ab.sh 3 "" "TCAR_ONEHOT_TIME=0" 2>&1 | tee gpurun_out/r3m_ab.txt
