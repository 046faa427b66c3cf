#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 bash tools/ab.sh 5 "" "TCAR_ONEHOT_TIME=0" "TCAR_FLAG_FORK=0" 2>&1 | tee gpurun_out/r3r_ab20.txt
