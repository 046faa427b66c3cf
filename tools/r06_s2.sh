#!/bin/bash
# round 6, GPU session 2: full GPU suite on the new tree, the sharded step sequenced in C++ (one rank), RCCL at world 1 in the bench,
# the 10 M-item configuration with the 256 x 384 dX tile / other dE tiles
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
S="bash tools/gpu_session.sh r06s2"
$S tests
Q="--no_cpu_baseline --no_e2e"
$S "bench:dp1_cxx:TCAR_FORCE_DP=1 --dp_mode sharded $Q" "bench:dp1_py:TCAR_FORCE_DP=1 TCAR_SHARD_PY_STEP=1 --dp_mode sharded $Q"
$S "bench:dp1_rccl:TCAR_FORCE_COLLECTIVES=1 --dp_mode sharded $Q" "bench:dp1_rccl_replica:TCAR_FORCE_COLLECTIVES=1 --dp_mode replica --scoring bf16x3 $Q"
$S "bench:sim8:TCAR_FORCE_DP=1 TCAR_SIM_WORLD=8 --dp_mode sharded $Q"
ST="--config stress10m --steps 10 --warmup 2 $Q"
$S "bench:stress_sk18:TCAR_SPLITK=18 $ST" "bench:stress_sk64:TCAR_SPLITK=64 $ST" "bench:stress_sk64_de256:TCAR_SPLITK=64 TCAR_BF16_TILE=2562 $ST" "bench:stress_sk64_dering:TCAR_SPLITK=64 TCAR_BF16_TILE=1923 $ST"
