#!/bin/bash
# sharded / replica DP code paths on one rank and as 2 ranks on one GPU (gloo dry run), device-sampler e2e, gather bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for mode in replica sharded; do
  TCAR_FORCE_DP=1 python bench.py --steps 200 --no_cpu_baseline --no_e2e --dp_mode $mode 2>gpurun_out/dp1_$mode.err | tail -1 > gpurun_out/bench_dp1_$mode.json
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_dp1_$mode.json")); print("$mode 1 rank:", d["ms_per_step"], d["value"], d.get("exchange",{}).get("mode"), d["host_enqueue_ms_per_step"])
PY
  python bench.py --gpus 2 --same_device --backend gloo --steps 30 --warmup 5 --no_cpu_baseline --dp_mode $mode 2>gpurun_out/dp2_$mode.err | tail -1 > gpurun_out/bench_dp2_$mode.json
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_dp2_$mode.json")); print("$mode 2 ranks (gloo, one GPU):", d["ms_per_step"], d["n_gpus"], json.dumps(d.get("exchange"))[:400])
PY
done
python bench.py > gpurun_out/bench_globo.json 2> gpurun_out/bench_globo.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_globo.json")); print("default:", d["ms_per_step"], d["value"], d["end_to_end_sessions_per_s"])
PY
