#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 bash tools/ab.sh 3 "" "TCAR_DBG_REST_WT=1" 2>&1 | tee gpurun_out/r3r_ab4.txt
