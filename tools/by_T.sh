#!/bin/bash
# ms_per_step_by_T of bench.py under each environment setting given as an argument ("" = default)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for s in "$@"; do
  out=$(env $s python bench.py --steps 50 --warmup 10 --no_cpu_baseline --no_e2e ${BY_T:+--by_T $BY_T} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_step_by_T'))")
  echo "[$s] $out" | tee -a gpurun_out/by_T.txt
done
