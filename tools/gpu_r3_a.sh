#!/bin/bash
# round 3, session A: full GPU suite, A/B of the split projections / one-shot ring, kernel trace + timeline of the default step
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=line 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r3a_pytest.log; cat gpurun_out/r3a_pytest.log
python bench.py --no_cpu_baseline > gpurun_out/r3a_bench_default.json 2> gpurun_out/r3a_bench_default.err; tail -c 1500 gpurun_out/r3a_bench_default.json
bash tools/ab.sh 2 "" "TCAR_PROJ_SPLIT=0" "TCAR_X3_ONESHOT=0" "TCAR_PROJ_SPLIT=0 TCAR_X3_ONESHOT=0" 2>&1 | tee gpurun_out/r3a_ab.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_r3a -o r3a -- python3 $OLDPWD/bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_e2e > $OLDPWD/gpurun_out/prof_r3a.log 2>&1 )
db=$(ls gpurun_out/prof_r3a/*/r3a_results.db gpurun_out/prof_r3a/r3a_results.db 2>/dev/null | head -1)
python tools/kstats.py $db gpurun_out/r3a_kstats.csv > gpurun_out/r3a_kstats.txt
python tools/timeline.py $db 100 > gpurun_out/r3a_timeline.txt
cat gpurun_out/r3a_timeline.txt
