cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python tools/anchor_grad_error.py 2 2>&1 | grep -v amdgpu.ids | tail -27 | cut -c1-120
timeout 2500 python -m pytest tests/test_gpu_parity.py -q -k "anchored or schedule_switches or ce_finish_as_one or fused_step_gradients" 2>&1 | grep -E "^E  +Assert|passed|failed|Error|^FAILED" | cut -c1-400
bash tools/ab_anchor.sh
