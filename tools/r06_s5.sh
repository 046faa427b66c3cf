#!/bin/bash
# round 6, GPU session 5: why is bench.py's forced-collective loop 1.8 ms per step when the same engine steps at 0.67 ms in tools/shard_host_profile.py?
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
Q="--dp_mode sharded --no_cpu_baseline --no_e2e --no_kernel_timing --steps 200 --warmup 20"
TCAR_FORCE_COLLECTIVES=1 python -m cProfile -s cumulative bench.py $Q 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | head -70 | cut -c1-200 > gpurun_out/r06s5_cprofile_bench_forced.txt
head -60 gpurun_out/r06s5_cprofile_bench_forced.txt
S="bash tools/gpu_session.sh r06s5"
$S "bench:forced_resident:TCAR_FORCE_COLLECTIVES=1 --resident_feed $Q" "bench:forced_sampler:TCAR_FORCE_COLLECTIVES=1 $Q"
$S "ab:4:TCAR_BF16_TILE=0|TCAR_BF16_TILE=1923"
