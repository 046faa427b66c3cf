#!/bin/bash
# round-2 evidence run: HR parity of the bench's default precision, default bench line, PMC passes of the scoring GEMMs as
# the default step runs them (fwd x3, dX / dE hi-only) and in bf16x3, kernel trace + timeline of the default bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_e2e.py -q -x -k "bf16_gradient" -s 2>&1 | tail -5
for m in "fwd 3" "dx 1" "de 1" "dx 3" "de 3"; do bash tools/pmc_gemm.sh $m > /dev/null 2>&1; done
ls gpurun_out/pmc_score_*.json
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 3000 gpurun_out/bench_default.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_i -o i -- python3 $OLDPWD/bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_e2e > $OLDPWD/gpurun_out/prof_i.log 2>&1 )
db=$(ls gpurun_out/prof_i/*/i_results.db gpurun_out/prof_i/i_results.db 2>/dev/null | head -1)
python tools/kstats.py $db gpurun_out/prof_i_kstats.csv > gpurun_out/prof_i_kstats.txt
python tools/timeline.py $db 100 > gpurun_out/prof_i_timeline.txt
tail -3 gpurun_out/prof_i_timeline.txt
