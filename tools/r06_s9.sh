#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
S="bash tools/gpu_session.sh r06s9"
Q="--dp_mode sharded --no_cpu_baseline --no_e2e --no_kernel_timing --steps 200 --warmup 20"
$S "tests:rccl or sharded or dp2"
for q in 4 8; do $S "bench:forced_q$q:GPU_MAX_HW_QUEUES=$q TCAR_FORCE_COLLECTIVES=1 $Q" "bench:forced_pg_q$q:GPU_MAX_HW_QUEUES=$q TCAR_FORCE_COLLECTIVES=1 TCAR_RCCL_DIRECT=0 $Q"; done
$S "bench:forced:TCAR_FORCE_COLLECTIVES=1 $Q" "bench:forced_replica:TCAR_FORCE_COLLECTIVES=1 --dp_mode replica --scoring bf16x3 --no_cpu_baseline --no_e2e --no_kernel_timing --steps 200 --warmup 20"
$S "bench:forced_full:TCAR_FORCE_COLLECTIVES=1 --dp_mode sharded --no_cpu_baseline --no_e2e"
