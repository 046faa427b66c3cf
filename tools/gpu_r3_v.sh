#!/bin/bash
# device-side timeline of the step: the host enqueues the 60 timed steps behind a 60-ms stall of the main stream
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_r3v -o r3v -- python3 $OLDPWD/bench.py --steps 60 --warmup 20 --stall_ms 60 --resident_feed --no_cpu_baseline --no_e2e --no_kernel_timing > $OLDPWD/gpurun_out/prof_r3v.log 2>&1 )
db=$(find gpurun_out/prof_r3v -name "*.db" | head -1)
[ -n "$db" ] && timeout 100 python tools/timeline.py $db 26 < /dev/null > gpurun_out/r3v_timeline.txt
cat gpurun_out/r3v_timeline.txt
grep '^{' gpurun_out/prof_r3v.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d['steps'])"
[ -n "$db" ] && timeout 60 python tools/kstats.py $db gpurun_out/r3v_kstats.csv < /dev/null | head -3
rm -rf gpurun_out/prof_r3v
