#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 bash tools/ab.sh 3 "" "TCAR_FLAG_FORK=0" 2>&1 | tail -6
