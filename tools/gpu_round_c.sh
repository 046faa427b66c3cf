#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python tools/x3_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/x3_bench.txt
for wg in 1 2; do TCAR_GATHER_WG=$wg python tools/gather_bench.py 2>&1 | grep -v amdgpu.ids | head -1; done | tee gpurun_out/gather_big.txt
TCAR_GATHER_BIG_ROWS=1000000000 python tools/gather_bench.py 2>&1 | grep -v amdgpu.ids | head -1 | tee -a gpurun_out/gather_big.txt
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -k "negative_modes or stress or throughput_form or 256x288 or de_tiles" 2>&1 | tail -8 | tee gpurun_out/pytest_c.log
