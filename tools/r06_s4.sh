#!/bin/bash
# round 6, GPU session 4: full suite on the pruned tree, host profile of the sharded step with live collectives, stress config with the
# catalog-size heuristics, default bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
S="bash tools/gpu_session.sh r06s4"
for m in direct pg none; do python tools/shard_host_profile.py $m 300 2>&1 | grep -v "amdgpu.ids\|Warning\|socket.cpp" > gpurun_out/r06s4_hostprof_$m.txt; head -45 gpurun_out/r06s4_hostprof_$m.txt | cut -c1-200; done
$S tests
$S "bench:stress:--config stress10m --steps 10 --warmup 2 --no_cpu_baseline --no_e2e"
$S bench:default
