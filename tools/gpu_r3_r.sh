#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python bench.py --gpus 2 --same_device --backend gloo --steps 10 --warmup 2 --no_cpu_baseline --dp_mode sharded 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], json.dumps(d['exchange'].get('ms_per_collective')))"
TCAR_FORCE_DP=1 timeout 200 python bench.py --no_cpu_baseline --no_e2e --dp_mode sharded --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], json.dumps(d['exchange'].get('ms_per_collective')))"
