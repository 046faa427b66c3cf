#!/bin/bash
# bench lines of the BASELINE configurations and precision modes (default flags otherwise), DP code paths on one rank
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for c in adressa mind; do
  python bench.py --config $c > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
done
python bench.py --config stress10m --steps 10 --warmup 2 --no_cpu_baseline --no_e2e > gpurun_out/bench_stress10m.json 2> gpurun_out/bench_stress10m.err
for s in bf16x3 bf16 f32; do
  python bench.py --scoring $s --no_cpu_baseline --no_e2e > gpurun_out/bench_globo_$s.json 2> gpurun_out/bench_globo_$s.err
done
for mode in replica sharded; do
  TCAR_FORCE_DP=1 python bench.py --no_cpu_baseline --no_e2e --dp_mode $mode 2>gpurun_out/dp1_$mode.err | tail -1 > gpurun_out/bench_dp1_$mode.json
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_*.json")):
    try:
        d = json.load(open(f))
        r = d.get("roofline") or {}
        print(f.split("/")[-1], d["ms_per_step"], d["value"], d.get("scoring"), (d.get("end_to_end_sessions_per_s") or {}).get("value"), r.get("tag"), r.get("frac"))
    except Exception as e:
        print(f, "unreadable", e)
PY
