#!/bin/bash
# round 6, GPU session 6: hardware-queue sharing — the forced-collective loop with the device sampler under GPU_MAX_HW_QUEUES
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
S="bash tools/gpu_session.sh r06s6"
Q="--dp_mode sharded --no_cpu_baseline --no_e2e --no_kernel_timing --steps 200 --warmup 20"
for q in 4 6 8 12; do
  $S "bench:forced_q$q:GPU_MAX_HW_QUEUES=$q TCAR_FORCE_COLLECTIVES=1 $Q" "bench:forced_pg_q$q:GPU_MAX_HW_QUEUES=$q TCAR_FORCE_COLLECTIVES=1 TCAR_RCCL_DIRECT=0 $Q"
done
for q in 4 8; do $S "bench:default_q$q:GPU_MAX_HW_QUEUES=$q --no_cpu_baseline --no_e2e --no_kernel_timing --steps 400 --warmup 20"; done
