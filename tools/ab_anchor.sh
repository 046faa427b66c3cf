#!/bin/bash
# A/B of the anchored softmax form (TCAR_FUSED_CE = 2, default) against the group-maximum form + rescale pass (1): interleaved rounds of
# the default bench on ONE box; results -> gpurun_out/r06_ab_anchor.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
mkdir -p gpurun_out
out=gpurun_out/r06_ab_anchor.txt; : > $out
ms() { python3 - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); print(d["ms_per_step"])
PY
}
for r in 1 2 3 4; do
  for f in 2 1; do
    TCAR_FUSED_CE=$f python bench.py --no_cpu_baseline --no_e2e --no_kernel_timing --no_by_T --steps 400 --warmup 20 > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err
    echo "round $r TCAR_FUSED_CE=$f ms/step $(ms gpurun_out/ab_tmp.json)" | tee -a $out
  done
done
