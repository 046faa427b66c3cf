#!/bin/bash
# same-box A/B of environment settings: every argument is one setting ("VAR=1 OTHER=0" or "" for the defaults); the default
# bench runs ROUNDS times per setting, interleaved.  Usage: tools/ab.sh 3 "" "TCAR_FUSED_Q=0"
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=$1; shift
for r in $(seq $ROUNDS); do
  for s in "$@"; do
    out=$(env $s python bench.py --steps 400 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "round $r [$s] $out"
  done
done
