#!/bin/bash
# sorted segmented item-row sum: host enqueue rate, kernel timeline of one step, per-kernel stats
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for s in "" "TCAR_SORT_SCATTER=0"; do
  env $s python bench.py --steps 400 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$s]', d['ms_per_step'], 'host enqueue', d['host_enqueue_ms_per_step'])"
done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_f -o f -- python3 $OLDPWD/bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing > $OLDPWD/gpurun_out/prof_f.log 2>&1 )
db=$(ls gpurun_out/prof_f/*/f_results.db gpurun_out/prof_f/f_results.db 2>/dev/null | head -1)
python tools/kstats.py $db gpurun_out/prof_f_kstats.csv | head -50
python tools/timeline.py $db 100
