#!/bin/bash
# round 3, end-of-round evidence: full GPU suite, smoke, default bench lines, configs / precisions / DP paths, kernel trace + timeline,
# PMC passes of the logits GEMM with its softmax epilogue
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=line 2>&1 | grep -v "^$" | tail -12 > gpurun_out/r3_pytest_gpu.log; cat gpurun_out/r3_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_default_20_5.json 2> /dev/null
python bench.py --resident_feed --no_cpu_baseline --no_e2e > gpurun_out/r3_bench_resident_feed.json 2> /dev/null
for c in adressa mind; do python bench.py --config $c > gpurun_out/r3_bench_$c.json 2> gpurun_out/r3_bench_$c.err; done
python bench.py --config stress10m --steps 10 --warmup 2 --no_cpu_baseline --no_e2e > gpurun_out/r3_bench_stress10m.json 2> gpurun_out/r3_bench_stress10m.err
for s in bf16x3 bf16 f32; do python bench.py --scoring $s --no_cpu_baseline --no_e2e > gpurun_out/r3_bench_globo_$s.json 2> /dev/null; done
for mode in replica sharded; do TCAR_FORCE_DP=1 python bench.py --no_cpu_baseline --no_e2e --dp_mode $mode 2>/dev/null | tail -1 > gpurun_out/r3_bench_dp1_$mode.json; done
for w in 2 4 8; do TCAR_FORCE_DP=1 TCAR_SIM_WORLD=$w python bench.py --no_cpu_baseline --no_e2e --dp_mode sharded 2>/dev/null | tail -1 > gpurun_out/r3_bench_simworld$w.json; done
for mode in replica sharded; do python bench.py --gpus 2 --same_device --backend gloo --steps 30 --warmup 5 --no_cpu_baseline --dp_mode $mode 2>/dev/null | tail -1 > gpurun_out/r3_bench_dp2_$mode.json; done
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_r3 -o fin -- python3 $OLDPWD/bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_e2e > $OLDPWD/gpurun_out/prof_r3.log 2>&1 )
db=$(ls gpurun_out/prof_r3/*/fin_results.db gpurun_out/prof_r3/fin_results.db 2>/dev/null | head -1)
python tools/kstats.py $db gpurun_out/r3_kernel_stats_default.csv > gpurun_out/r3_kstats.txt
python tools/timeline.py $db 100 > gpurun_out/r3_timeline_one_step.txt
bash tools/pmc_gemm.sh fwdce2 3 > gpurun_out/r3_pmc_fwdce.log 2>&1
python tools/head_bench.py 2 2>&1 | grep -v Warning > gpurun_out/r3_head_bench.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3_bench_*.json")):
    try:
        d = json.load(open(f))
        r = d.get("roofline") or {}
        print(f.split("/")[-1], d["ms_per_step"], d["value"], d.get("device_step_sessions_per_s"), d.get("scoring"), (d.get("end_to_end_sessions_per_s") or {}).get("value"), r.get("tag"), r.get("bound"), r.get("frac"), r.get("traffic"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -2 gpurun_out/r3_timeline_one_step.txt; cat gpurun_out/pmc_score_fwd_ce_n3.json 2>/dev/null | head -30
