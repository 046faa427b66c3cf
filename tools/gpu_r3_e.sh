#!/bin/bash
# round 3, session E: sampler-in-loop bench, planned-schedule tests, A/B of the post-dX chain placement
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_sampler.py tests/test_gpu_e2e.py tests/test_gpu_sharded.py tests/test_gpu_dp2.py -m gpu -q --tb=short -x 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r3e_pytest.log; cat gpurun_out/r3e_pytest.log
python bench.py > gpurun_out/r3e_bench_default.json 2> gpurun_out/r3e_bench_default.err; tail -3 gpurun_out/r3e_bench_default.err; python - <<PY
import json
d = json.load(open("gpurun_out/r3e_bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step", "device_step_sessions_per_s", "timed_loop", "host_enqueue_ms_per_step")})
print(d["cpu_baseline"]); print(d["end_to_end_sessions_per_s"]); print(d["roofline"]["tag"], d["roofline"]["frac"], d["roofline"]["avg_ms"])
PY
bash tools/ab.sh 2 "" "TCAR_CHAIN_AFTER_DE=1" 2>&1 | tee gpurun_out/r3e_ab.txt
( cd /tmp && export TMPDIR=/tmp && TCAR_CHAIN_AFTER_DE=1 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_r3e -o r3e -- python3 $OLDPWD/bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_e2e > $OLDPWD/gpurun_out/prof_r3e.log 2>&1 )
db=$(ls gpurun_out/prof_r3e/*/r3e_results.db gpurun_out/prof_r3e/r3e_results.db 2>/dev/null | head -1)
python tools/timeline.py $db 60 > gpurun_out/r3e_timeline_chain_after_de.txt
cat gpurun_out/r3e_timeline_chain_after_de.txt
