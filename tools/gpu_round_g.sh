#!/bin/bash
# bf16x3-mixed (bf16x3 forward, plain-bf16 gradient GEMMs): step time, timeline, kernel stats
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for s in bf16x3 bf16x3-mixed bf16; do
  python bench.py --scoring $s --steps 400 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$s]', d['ms_per_step'], 'host enqueue', d['host_enqueue_ms_per_step'])"
done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_g -o g -- python3 $OLDPWD/bench.py --scoring bf16x3-mixed --steps 200 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing > $OLDPWD/gpurun_out/prof_g.log 2>&1 )
db=$(ls gpurun_out/prof_g/*/g_results.db gpurun_out/prof_g/g_results.db 2>/dev/null | head -1)
python tools/kstats.py $db gpurun_out/prof_g_kstats.csv | head -24
python tools/timeline.py $db 100
