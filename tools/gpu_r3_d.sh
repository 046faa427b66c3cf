#!/bin/bash
# round 3, session D: transposed softmax epilogue (tests, A/B, timeline), head micro-benchmark on the right stream
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q --tb=short -x -k "softmax_epilogue or globo_full_size or (step_matches_oracle and mixed) or mixed_precision_backward or deferred or same_step_twice" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r3d_pytest.log; cat gpurun_out/r3d_pytest.log
python tools/head_bench.py 2 2>&1 | grep -v Warning | tee gpurun_out/r3d_head_bench.txt
bash tools/ab.sh 2 "" "TCAR_FUSED_CE=0" "TCAR_PROJ_SPLIT=0 TCAR_X3_ONESHOT=0" 2>&1 | tee gpurun_out/r3d_ab.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_r3d -o r3d -- python3 $OLDPWD/bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_e2e > $OLDPWD/gpurun_out/prof_r3d.log 2>&1 )
db=$(ls gpurun_out/prof_r3d/*/r3d_results.db gpurun_out/prof_r3d/r3d_results.db 2>/dev/null | head -1)
python tools/timeline.py $db 60 > gpurun_out/r3d_timeline.txt
cat gpurun_out/r3d_timeline.txt
