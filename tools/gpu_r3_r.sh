#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 bash tools/ab.sh 6 "" "TCAR_FLAG_FORK=242" 2>&1 | tee gpurun_out/r3r_ab7.txt
