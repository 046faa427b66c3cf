#!/bin/bash
# round 3, session B: fixed tests, A/B of the update placement, timeline of the non-deferred step
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sampler.py tests/test_gpu_sharded.py tests/test_gpu_configs.py tests/test_gpu_e2e.py -m gpu -q --tb=line -k "deferred or negative_modes_follow or two_ranks_sharded or globo_full_size or oracle_run or (step_matches_oracle and mixed) or (negative_modes and mixed)" 2>&1 | grep -v "^$" | tail -15 > gpurun_out/r3b_pytest.log; cat gpurun_out/r3b_pytest.log
bash tools/ab.sh 2 "" "TCAR_REST_EARLY=0" "TCAR_NO_DEFER=1" "TCAR_REST_GRID=256" "TCAR_REST_EARLY=0 TCAR_PROJ_SPLIT=0 TCAR_X3_ONESHOT=0" 2>&1 | tee gpurun_out/r3b_ab.txt
( cd /tmp && export TMPDIR=/tmp && TCAR_NO_DEFER=1 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_r3b -o r3b -- python3 $OLDPWD/bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_e2e > $OLDPWD/gpurun_out/prof_r3b.log 2>&1 )
db=$(ls gpurun_out/prof_r3b/*/r3b_results.db gpurun_out/prof_r3b/r3b_results.db 2>/dev/null | head -1)
python tools/timeline.py $db 60 > gpurun_out/r3b_timeline_nodefer.txt
cat gpurun_out/r3b_timeline_nodefer.txt
