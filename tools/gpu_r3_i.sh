#!/bin/bash
# round 3, session I: non-temporal streaming of the Adam rest pass / dE result (A/B), bench check
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TCAR_NT=3 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "deferred or golden or globo_full_size" 2>&1 | grep -v "^$" | tail -4
bash tools/ab.sh 3 "" "TCAR_NT=1" "TCAR_NT=2" "TCAR_NT=3" 2>&1 | tee gpurun_out/r3i_ab.txt
python bench.py --no_cpu_baseline --no_e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_with_kernel_events'], d['roofline']['traffic'], d['roofline']['frac'])"
