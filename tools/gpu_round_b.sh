#!/bin/bash
# GPU pass B: full -m gpu suite (no -x: list every failure), default bench, kernel trace + timeline, GEMM micro-benchmarks
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
tail -15 gpurun_out/pytest.log
python bench.py --no_e2e --no_cpu_baseline > gpurun_out/bench_globo.json 2> gpurun_out/bench_globo.err; tail -c 300 gpurun_out/bench_globo.json
for m in fwd dx de both; do python tools/gemm_bench.py $m 3 20; done > gpurun_out/gemm_alone.txt 2>&1
GB_SPLITK=42 python tools/gemm_bench.py dx 3 20 >> gpurun_out/gemm_alone.txt 2>&1
GB_SPLITK=42 python tools/gemm_bench.py both 3 20 >> gpurun_out/gemm_alone.txt 2>&1
TCAR_TILE288=0 GB_SPLITK=36 python tools/gemm_bench.py both 3 20 >> gpurun_out/gemm_alone.txt 2>&1
grep -v amdgpu.ids gpurun_out/gemm_alone.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_b -o b -- python3 $OLDPWD/bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_e2e > $OLDPWD/gpurun_out/prof_b.log 2>&1 )
DB=$(find gpurun_out/prof_b -name "*.db" | head -1)
python tools/kstats.py $DB gpurun_out/r02_b_kernel_stats.csv > gpurun_out/r02_b_kstats.txt 2>&1
python tools/timeline.py $DB 100 > gpurun_out/r02_b_timeline.txt 2>&1
cat gpurun_out/r02_b_timeline.txt
