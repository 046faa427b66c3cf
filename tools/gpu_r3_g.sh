#!/bin/bash
# round 3, session G: late fork of the Adam rest pass (A/B + timelines)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "deferred or golden" 2>&1 | grep -v "^$" | tail -5
TCAR_REST_AFTER=2 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py -m gpu -q --tb=short -x -k "deferred or bit_identical" 2>&1 | grep -v "^$" | tail -5
bash tools/ab.sh 3 "" "TCAR_REST_AFTER=1" "TCAR_REST_AFTER=2" 2>&1 | tee gpurun_out/r3g_ab.txt
for v in 1 2; do
( cd /tmp && export TMPDIR=/tmp && TCAR_REST_AFTER=$v rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_r3g$v -o r3g -- python3 $OLDPWD/bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_e2e --resident_feed > $OLDPWD/gpurun_out/prof_r3g$v.log 2>&1 )
db=$(ls gpurun_out/prof_r3g$v/*/r3g_results.db gpurun_out/prof_r3g$v/r3g_results.db 2>/dev/null | head -1)
python tools/timeline.py $db 60 > gpurun_out/r3g_timeline_rest_after_$v.txt
head -14 gpurun_out/r3g_timeline_rest_after_$v.txt; tail -2 gpurun_out/r3g_timeline_rest_after_$v.txt
done
