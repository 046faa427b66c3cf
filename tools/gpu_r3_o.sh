#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --kernel-trace -d /tmp/prof_ev -o p -- $OLDPWD/tools/micro/event_cost > /tmp/ev.log 2>&1 ; tail -2 /tmp/ev.log )
find /tmp/prof_ev -type f | head
db=$(find /tmp/prof_ev -name "*.db" | head -1)
[ -n "$db" ] && timeout 60 python tools/micro/event_cost.py $db < /dev/null 2>&1 | tee gpurun_out/r3o_event_cost.txt
