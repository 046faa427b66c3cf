#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
S="bash tools/gpu_session.sh r06s10"
$S "tests:flag_and_event or fork or two_engines or same_step_twice or bit_identical"
$S "ab:5:TCAR_FLAG_FORK=4095|TCAR_FLAG_FORK=3839"
