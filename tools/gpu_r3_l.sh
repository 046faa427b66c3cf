#!/bin/bash
# one-hot time contraction: op-level tests, full-size parity, in-step kernel durations, A/B of the step
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "softmax_epilogue or time_onehot or full_size or step_matches_oracle or deferred" 2>&1 | tail -15
run() {
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && export "$@" && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python3 $OLDPWD/bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing > /tmp/prof_$tag.log 2>&1 )
  db=$(ls /tmp/prof_$tag/*/p_results.db /tmp/prof_$tag/p_results.db 2>/dev/null | head -1)
  echo "== $tag $db"
  [ -n "$db" ] && timeout 60 python tools/kstats.py $db /tmp/ks_$tag.csv < /dev/null | grep -E "gemm_bf16_kernel<0, 0, 3|time_scores" | cut -c1-140
}
run onehot X=1
run classic TCAR_ONEHOT_TIME=0
timeout 600 bash tools/ab.sh 3 "" "TCAR_ONEHOT_TIME=0" 2>&1 | tee gpurun_out/r3l_ab.txt
