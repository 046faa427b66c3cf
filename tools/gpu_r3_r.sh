#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for mode in replica sharded; do TCAR_FORCE_DP=1 timeout 200 python bench.py --no_cpu_baseline --no_e2e --dp_mode $mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'])"; done
TCAR_FORCE_DP=1 TCAR_SIM_WORLD=8 timeout 200 python bench.py --no_cpu_baseline --no_e2e --dp_mode sharded 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('simworld8', d['ms_per_step'])"
timeout 200 python tools/eval_bench.py 2>&1 | tail -2
