#!/bin/bash
# round 6, GPU session 3: RCCL called directly (tests + bench, against the process-group path), the pruned switch set's tests,
# the erratum bisect, dE at the 10 M-item scale (HBM bytes under the counters + tile variants)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
S="bash tools/gpu_session.sh r06s3"
$S "tests:rccl or schedule_switches or onehot_gradient or sharded or dp2"
Q="--no_cpu_baseline --no_e2e"
$S "bench:dp1_rccl_direct:TCAR_FORCE_COLLECTIVES=1 --dp_mode sharded $Q" "bench:dp1_rccl_pg:TCAR_FORCE_COLLECTIVES=1 TCAR_RCCL_DIRECT=0 --dp_mode sharded $Q"
$S "bench:dp1_rccl_replica_direct:TCAR_FORCE_COLLECTIVES=1 --dp_mode replica --scoring bf16x3 $Q" "bench:dp1_rccl_replica_pg:TCAR_FORCE_COLLECTIVES=1 TCAR_RCCL_DIRECT=0 --dp_mode replica --scoring bf16x3 $Q"
( cd tools/micro && timeout 600 ./pkfma_lds_slp 2 5 2 ) 2>&1 | tee gpurun_out/r06s3_erratum_bisect.txt | tail -30
gb() { echo -n "$*: "; env "${@:1:$#-1}" python tools/gemm_bench.py ${!#} 1 5 2>&1 | grep -v Warning | tail -1; }
for t in 0 2562 1923 128 1283; do gb GB_N=10000000 GB_TILE=$t de2; done 2>&1 | tee gpurun_out/r06s3_de_10m.txt
GB_N=10000000 bash tools/pmc_gemm.sh de2 1 3 > gpurun_out/r06s3_pmc_de2_10m.log 2>&1; tail -3 gpurun_out/r06s3_pmc_de2_10m.log; cat gpurun_out/pmc_de2_n1.json 2>/dev/null | head -40
