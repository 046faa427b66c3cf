#!/bin/bash
# One parameterised GPU-box session (replaces the per-session gpu_r*_*.sh scripts).  Every step writes under gpurun_out/<tag>_*.
#   tools/gpu_session.sh <tag> <step> [<step> ...]
# steps:
#   tests[:<pytest -k expr>]   pytest -m gpu (optionally filtered)        -> <tag>_pytest.log
#   smoke                      __graft_entry__.smoke()
#   bench[:<name>[:<bench.py args>]]   one bench line                     -> <tag>_bench_<name>.json
#                              (args may start with environment assignments: "bench:dp1:TCAR_FORCE_DP=1 --dp_mode sharded")
#   ab:<rounds>:<env A>|<env B>|...    interleaved A/B of environment settings (headline loop only), ms/step per run
#   trace[:<bench.py args>]    rocprofv3 --kernel-trace of the default bench -> <tag>_kernel_stats.csv/.txt, <tag>_timeline.txt
#   pmc:<variant>:<nsplit>     tools/pmc_gemm.sh                          -> <tag>_pmc_<variant>.log (+ gpurun_out/pmc_*.json)
#   head                       tools/head_bench.py                        -> <tag>_head_bench.txt
#   py:<script and args>       any python tool                            -> <tag>_py.log (appended)
#   sh:<command line>          any shell command (e.g. a tools/micro binary)  -> <tag>_sh.log (appended)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
tag=$1; shift
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" == "$step" ] && rest=""
  echo "=== [$tag] $step"
  case $kind in
    tests)
      if [ -n "$rest" ]; then timeout 1700 python -m pytest tests -m gpu -x -q --tb=short --timeout=300 -k "$rest" > gpurun_out/${tag}_pytest.log 2>&1
      else timeout 1700 python -m pytest tests -m gpu -x -q --tb=short --timeout=300 > gpurun_out/${tag}_pytest.log 2>&1; fi
      tail -25 gpurun_out/${tag}_pytest.log | cut -c1-400 ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    bench)
      name=${rest%%:*}; args=${rest#*:}; [ "$args" == "$rest" ] && args=""; [ -z "$name" ] && name=default
      envs=""; while [[ "$args" =~ ^([A-Z_0-9]+=[^ ]*)\ ?(.*)$ ]]; do envs="$envs ${BASH_REMATCH[1]}"; args="${BASH_REMATCH[2]}"; done
      env $envs python bench.py $args > gpurun_out/${tag}_bench_${name}.json 2> gpurun_out/${tag}_bench_${name}.err
      python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${tag}_bench_${name}.json").read().splitlines() if l.startswith("{")][-1])
    r = d.get("roofline") or {}
    print("${name}", "ms/step", d["ms_per_step"], "sessions/s", d["value"], "device-step", d.get("device_step_sessions_per_s"),
          "e2e", (d.get("end_to_end_sessions_per_s") or {}).get("value"), "roofline", r.get("tag"), r.get("frac"),
          "kernels", {k: v["avg_ms"] for k, v in (d.get("kernels") or {}).items()})
except Exception as e:
    print("bench ${name}: unreadable", e); print(open("gpurun_out/${tag}_bench_${name}.err").read()[-1500:])
PY
      ;;
    ab)
      rounds=${rest%%:*}; sets=${rest#*:}
      IFS='|' read -ra S <<< "$sets"
      for r in $(seq $rounds); do for s in "${S[@]}"; do
        out=$(env $s python bench.py --steps 400 --warmup 20 --no_cpu_baseline --no_e2e --no_kernel_timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
        echo "round $r [$s] $out" | tee -a gpurun_out/${tag}_ab.txt
      done; done ;;
    trace)
      envs=""; while [[ "$rest" =~ ^([A-Z_0-9]+=[^ ]*)\ ?(.*)$ ]]; do envs="$envs ${BASH_REMATCH[1]}"; rest="${BASH_REMATCH[2]}"; done
      ( cd /tmp && export TMPDIR=/tmp $envs && timeout 400 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_${tag} -o tr -- python3 $OLDPWD/bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_e2e --no_by_T $rest > $OLDPWD/gpurun_out/${tag}_trace.log 2>&1 )
      db=$(ls gpurun_out/prof_${tag}/*/tr_results.db gpurun_out/prof_${tag}/tr_results.db 2>/dev/null | head -1)
      python tools/kstats.py $db gpurun_out/${tag}_kernel_stats.csv > gpurun_out/${tag}_kernel_stats.txt
      python tools/timeline.py $db ${TL_STEP:-100} > gpurun_out/${tag}_timeline.txt
      cat gpurun_out/${tag}_timeline.txt; head -45 gpurun_out/${tag}_kernel_stats.txt
      rm -rf gpurun_out/prof_${tag} ;;
    pmc)
      v=${rest%%:*}; n=${rest#*:}
      bash tools/pmc_gemm.sh $v $n > gpurun_out/${tag}_pmc_${v}.log 2>&1; tail -5 gpurun_out/${tag}_pmc_${v}.log ;;
    head) python tools/head_bench.py 2 2>&1 | grep -v Warning | tee gpurun_out/${tag}_head_bench.txt ;;
    py) python $rest 2>&1 | tee -a gpurun_out/${tag}_py.log | tail -40 ;;
    sh) timeout 600 bash -c "$rest" 2>&1 | tee -a gpurun_out/${tag}_sh.log | tail -60 ;;
    *) echo "unknown step $step" ;;
  esac
done
