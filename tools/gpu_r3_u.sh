#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -v "^$" | tail -15 > gpurun_out/r3u_pytest_gpu.log; cat gpurun_out/r3u_pytest_gpu.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/r3u_bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/r3u_bench_default.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('host_enqueue_ms_per_step'))"
