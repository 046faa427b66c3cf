#!/bin/bash
# in-step duration of the logits GEMM as a function of its K (timing only; the debug settings give wrong numbers)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
run() {
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && export "$@" && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python3 $OLDPWD/bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing > /tmp/prof_$tag.log 2>&1 )
  db=$(ls /tmp/prof_$tag/*/p_results.db /tmp/prof_$tag/p_results.db 2>/dev/null | head -1)
  echo "== $tag $db"
  [ -n "$db" ] && timeout 60 python tools/kstats.py $db /tmp/ks_$tag.csv < /dev/null | grep -E "gemm_bf16_kernel<0, 0, 3|time_scores" | cut -c1-140
}
run onehot X=1
run noseg2 TCAR_DBG_NOSEG2=1
run classic TCAR_ONEHOT_TIME=0
run classic672 TCAR_ONEHOT_TIME=0 TCAR_DBG_K=672
run classic512 TCAR_ONEHOT_TIME=0 TCAR_DBG_K=512
