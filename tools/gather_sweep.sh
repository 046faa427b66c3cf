#!/bin/bash
# Forms of the embedding gather's throughput kernel at 655,360 rows (tools/gather_bench.py), interleaved: TCAR_GATHER_WG = workgroups
# per CU + 16 * form (0 = shipped: 2 rows per wave and trip, non-temporal stores; 1 plain stores (4 rows); 2 eight rows; 3 four rows; 4 one row).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq ${2:-3}); do for v in ${1:-2 1 18 34 50 66}; do
  echo -n "TCAR_GATHER_WG=$v  "; TCAR_GATHER_WG=$v python tools/gather_bench.py 2>&1 | grep "gather_clip_fwd:" 
done; done
