#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
S="bash tools/gpu_session.sh r06s8"
Q="--dp_mode sharded --no_cpu_baseline --no_e2e --no_kernel_timing --steps 200 --warmup 20"
$S "bench:forced:TCAR_FORCE_COLLECTIVES=1 $Q" "bench:forced_third:TCAR_SAMPLER_STREAM=third TCAR_FORCE_COLLECTIVES=1 $Q" "bench:forced_aux:TCAR_SAMPLER_STREAM=aux TCAR_FORCE_COLLECTIVES=1 $Q" "bench:forced_noprio:TCAR_NO_PRIO=1 TCAR_FORCE_COLLECTIVES=1 $Q" "bench:forced_noflag:TCAR_NO_FLAG_FORK=1 TCAR_FORCE_COLLECTIVES=1 $Q" "bench:forced_chunk1:TCAR_FEED_CHUNK=1 TCAR_FORCE_COLLECTIVES=1 $Q"
# one traced run of the slow configuration: where does the step's time go on the device?
( cd /tmp && export TMPDIR=/tmp TCAR_FORCE_COLLECTIVES=1 && timeout 400 rocprofv3 --kernel-trace -d $OLDPWD/gpurun_out/prof_forced -o tr -- python3 $OLDPWD/bench.py $Q > $OLDPWD/gpurun_out/r06s8_trace.log 2>&1 )
db=$(ls gpurun_out/prof_forced/*/tr_results.db gpurun_out/prof_forced/tr_results.db 2>/dev/null | head -1)
python tools/timeline.py $db 100 > gpurun_out/r06s8_timeline_forced.txt; cat gpurun_out/r06s8_timeline_forced.txt | cut -c1-140
rm -rf gpurun_out/prof_forced
