cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in adressa mind; do for r in 1 2; do for sk in 36 18 12 24; do
  out=$(TCAR_SPLITK=$sk python bench.py --config $cfg --steps 400 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$cfg round $r splitk=$sk $out"
done; done; done
for r in 1 2; do for sk in 36 18; do
  out=$(TCAR_SPLITK=$sk TCAR_FORCE_DP=1 python bench.py --dp_mode sharded --steps 400 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "sharded1 round $r splitk=$sk $out"
done; done
