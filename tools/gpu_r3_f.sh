#!/bin/bash
# round 3, session F: sort tests, negative rows on the third stream (A/B), gather PMC, timeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q --tb=short -x -k "sorted_segmented or same_step_twice or globo_full_size or gemm_grouped" 2>&1 | grep -v "^$" | tail -20 > gpurun_out/r3f_pytest.log; cat gpurun_out/r3f_pytest.log
bash tools/ab.sh 3 "" "TCAR_NEG_S3=0" 2>&1 | tee gpurun_out/r3f_ab.txt
bash tools/pmc_gather.sh 2>&1 | tail -25
python tools/gather_bench.py 2>&1 | grep -v Warn | head -5 | tee gpurun_out/r3f_gather_bench.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_r3f -o r3f -- python3 $OLDPWD/bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_e2e --resident_feed > $OLDPWD/gpurun_out/prof_r3f.log 2>&1 )
db=$(ls gpurun_out/prof_r3f/*/r3f_results.db gpurun_out/prof_r3f/r3f_results.db 2>/dev/null | head -1)
python tools/timeline.py $db 60 > gpurun_out/r3f_timeline.txt
cat gpurun_out/r3f_timeline.txt
