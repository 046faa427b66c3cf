"""CPU test of bench.py's rank launch: a plain `python bench.py --gpus N` (no torch.distributed.run around it) must start
its ranks as CHILD processes, relay rank 0's JSON line and exit 0 — the parent never initialises a GPU (bench.py:launch_ranks)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, env=e, timeout=240)


def test_plain_invocation_with_gpus_2_launches_two_ranks():
    r = _run(["--gpus", "2", "--launch_check", "--backend", "gloo"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # exactly one JSON line on stdout
    d = json.loads(lines[0])
    assert d["metric"] == "launch_check" and d["n_gpus"] == 2
    # dp.preflight ran the three collectives of the exchanges on every rank (gloo has no reduce-scatter: reported, not fatal)
    assert d["collectives"]["world"] == 2 and d["collectives"]["backend"] == "gloo"
    assert "collective preflight ok" in r.stderr


def test_plain_invocation_with_gpus_8_launches_eight_ranks():
    # the node shape of BASELINE.json's metric ("at 1/2/4/8 MI355X"): eight ranks rendezvous on 127.0.0.1 and run the preflight
    r = _run(["--gpus", "8", "--launch_check", "--backend", "gloo"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"] == "launch_check" and d["n_gpus"] == 8 and d["collectives"]["world"] == 8
    assert r.stderr.count("collective preflight ok") >= 1


def test_under_torch_distributed_run_it_is_a_rank_not_a_launcher():
    # the driver's own launch form: bench.py must not spawn again when WORLD_SIZE is already set
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch_check", "--backend", "gloo"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_failed_ranks_give_a_nonzero_exit_code():
    # no GPU here: the real bench ranks fail at torch.cuda.set_device; the launcher must pass the failure on
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a machine without a GPU")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no_cpu_baseline"])
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_launch_check_on_the_rccl_backend_fails_loudly_without_gpus():
    # `--launch_check` with the default backend (nccl = RCCL) puts rank r on cuda:r: on a box without GPUs every rank must
    # fail with a non-zero exit code (never hang, never fall back to CPU)
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a machine without a GPU")
    r = _run(["--gpus", "2", "--launch_check"])
    assert r.returncode != 0
