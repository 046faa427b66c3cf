#!/bin/bash
# round-2 GPU pass A: full -m gpu suite, bench lines of three configs, the self-launched 2-rank dry run, kernel trace, PMC of dE / dX
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
tail -5 gpurun_out/pytest.log
python bench.py > gpurun_out/bench_globo.json 2> gpurun_out/bench_globo.err; tail -c 600 gpurun_out/bench_globo.json
python bench.py --config adressa > gpurun_out/bench_adressa.json 2> gpurun_out/bench_adressa.err
python bench.py --config mind > gpurun_out/bench_mind.json 2> gpurun_out/bench_mind.err
python bench.py --gpus 2 --same_device --backend gloo --steps 50 --no_cpu_baseline --dp_mode replica > gpurun_out/bench_dp2_dry.json 2> gpurun_out/bench_dp2_dry.err; echo "dp2 rc=$?"
for m in fwd dx de both; do python tools/gemm_bench.py $m 3 20; done > gpurun_out/gemm_alone.txt 2>&1
cat gpurun_out/gemm_alone.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_a -o a -- python3 $OLDPWD/bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_e2e > $OLDPWD/gpurun_out/prof_a.log 2>&1 )
DB=$(find gpurun_out/prof_a -name "*.db" | head -1)
python tools/kstats.py $DB gpurun_out/r02_a_kernel_stats.csv > gpurun_out/r02_a_kstats.txt 2>&1
python tools/timeline.py $DB 100 > gpurun_out/r02_a_timeline.txt 2>&1
rm -rf gpurun_out/prof_a
tools/pmc_gemm.sh de 5; tools/pmc_gemm.sh dx 5
ls gpurun_out
