#!/bin/bash
# kernel timeline of the end-to-end trainer loop (device sampler)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OLDPWD/gpurun_out/prof_m -o m -- python3 $OLDPWD/main.py --synthetic 46033 --synthetic_train 150000 --synthetic_test 2000 --epoch 2 --gap_mode click_delta --scoring bf16x3-mixed --device_sampler ${1:-1} > $OLDPWD/gpurun_out/prof_m.log 2>&1 )
tail -3 gpurun_out/prof_m.log
db=$(ls gpurun_out/prof_m/*/m_results.db gpurun_out/prof_m/m_results.db 2>/dev/null | head -1)
python tools/kstats.py $db gpurun_out/prof_m_kstats.csv | head -45
python tools/timeline.py $db 300 | head -70
