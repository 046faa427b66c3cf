#!/bin/bash
# host-side cost of a step: HIP runtime API statistics (counts and mean durations per call)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for s in "X=1" "TCAR_FLAG_FORK=0"; do
  tag=$(echo $s | tr '=' '_')
  ( cd /tmp && export TMPDIR=/tmp && export $s && timeout 300 rocprofv3 --hip-runtime-trace --stats -d /tmp/prof_h_$tag -o h -- python3 $OLDPWD/bench.py --steps 400 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing > /tmp/prof_h_$tag.log 2>/tmp/prof_h_$tag.err )
  grep '^{' /tmp/prof_h_$tag.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$s', d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
  db=$(find /tmp/prof_h_$tag -name "*.db" | head -1)
  echo "== $s $db"
  timeout 60 python tools/api_stats.py $db < /dev/null
done 2>&1 | tee gpurun_out/r3q_host_api.txt
