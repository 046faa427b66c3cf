cd ${GRAFT_REPO_ROOT:-/root/repo}; python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
Q="--dp_mode sharded --no_cpu_baseline --no_e2e --no_kernel_timing --steps 400 --warmup 20"
for r in 1 2 3 4; do for m in 4095 3839; do
  v=$(TCAR_FORCE_DP=1 TCAR_FLAG_FORK=$m python bench.py $Q 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  w=$(TCAR_FORCE_DP=1 TCAR_SIM_WORLD=8 TCAR_FLAG_FORK=$m python bench.py $Q 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "round $r mask $m one-rank $v simworld8 $w" | tee -a gpurun_out/r06_ab_shard_startflag.txt
done; done
