#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 bash tools/ab.sh 5 "" "TCAR_DBG_RD=0" 2>&1 | tee gpurun_out/r3r_ab13.txt
