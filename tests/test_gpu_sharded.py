"""Catalog-sharded data-parallel step (tcar_amd.sharded.ShardedEngine).

* world 1: the Python-sequenced sharded path (per-shard softmax statistics, combine, range scatters, shard ctx update) against
  the fp64 oracle — loss, every gradient, the clip norms, the variables after Adam steps;
* world 2 on ONE GPU (gloo moving CUDA tensors through the host; RCCL needs > 1 GPU): uneven session shards and a rank whose
  shard of a batch is empty, against a single engine that sees the whole batch; the replicas must agree BIT FOR BIT (every item
  row has one owner, the dense-weight norms are broadcast)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("K", [7, 0])
def test_single_rank_sharded_path_matches_oracle(K):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tcar_amd  # noqa: F401
    from oracle.tcar_oracle import TcarOracle
    from tcar_amd.sharded import ShardedEngine
    from test_gpu_parity import _case, close
    N, H, Ht, B, T = 1000, 250, 64, 33, 5
    params, content, mw, batch = _case(N, H, Ht, B, T, max(K, 1), seed=321)
    if K == 0:
        batch = {k: v for k, v in batch.items() if k != "neg"}
    eng = ShardedEngine(params, content, mw, max_grad=2.0, scoring="bf16x3", world=1, rank=0)
    ora = TcarOracle(params, content, mw, max_grad=2.0)
    loss = eng.loss_and_grads(batch, cap=B + 3)                      # padded session capacity, as an uneven shard has
    o, g_o, sq_o = ora.loss_and_grads(batch)
    close(loss.cpu().numpy(), o["loss"].detach().numpy(), name="loss")
    g_e, sq_e = eng.export_grads(), eng.export_sqnorms()
    for k in g_o:
        close(g_e[k], g_o[k].numpy(), name="grad " + k, atol_scale=5e-5)
        assert abs(sq_e[k] - sq_o[k]) <= 2e-3 * sq_o[k] + 1e-12, ("sqnorm", k, sq_e[k], sq_o[k])
    for _ in range(2):
        close(eng.train_step(batch).cpu().numpy(), ora.train_step(batch).numpy(), name="train loss")
    p_e, p_o = eng.export_params(), ora.export()
    for k in p_o:      # two Adam steps at lr 1e-3: a rounding-level gradient coordinate may move by up to ~2e-3 * |w| scale
        close(p_e[k], p_o[k], name="param " + k, atol_scale=1e-3)
    rank, topk, ce, logits = eng.eval_step(batch, keep_logits=True)
    lo, ce_o = ora.eval_batch(batch)
    close(logits.cpu().numpy(), lo.numpy(), name="eval logits", atol_scale=1e-4)
    close(ce.cpu().numpy(), ce_o.numpy(), name="eval ce")


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tcar_amd  # noqa: F401
        from tcar_amd.dp import shard_bounds
        from tcar_amd.engine import TcarEngine
        from tcar_amd.sharded import ShardedEngine
        from test_gpu_parity import _case
        N, H, Ht, B, T, K = 1000, 250, 64, 37, 4, 6          # 37 sessions: uneven shards (19 + 18)
        params, content, mw, batch = _case(N, H, Ht, B, T, K, seed=5)
        _, _, _, tiny = _case(N, H, Ht, 1, 2, K, seed=6)      # a batch of ONE session: rank 1's shard is empty
        eng = ShardedEngine(params, content, mw, max_grad=2.0, group=dist.group.WORLD, scoring="bf16x3")
        assert eng.S == 512 and eng.nl == (512 if rank == 0 else 488)
        for step in range(4):
            full = tiny if step == 2 else batch
            b, t = full["seq"].shape
            lo, hi, cap = shard_bounds(b, world, rank)
            sub = {k: v[lo:hi] for k, v in full.items()} if hi > lo else None
            eng.train_step(sub, cap=cap, T=t, K=K)
        torch.cuda.synchronize()
        got = eng.export_params()
        if rank == 0:
            ref = TcarEngine(params, content, mw, max_grad=2.0, scoring="bf16x3")
            for step in range(4):
                ref.train_step(tiny if step == 2 else batch)
            want = ref.export_params()
            for k in want:
                d = np.abs(got[k] - want[k]).max()
                # same kernels, different summation order (float atomics, reduction trees): Adam bound as in test_gpu_parity
                assert d <= 1e-3 * np.abs(want[k]).max() + 0.25 * 1e-3 * 4, (k, d)
        flat = torch.cat([torch.tensor(v).reshape(-1) for v in got.values()])
        other = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        assert torch.equal(other[0], other[1]), float((other[0] - other[1]).abs().max())     # replicas: bit for bit
        info = eng.exchange_info()
        assert info["mode"] == "sharded" and info["bytes_per_step"]["item_rows"] == 4 * 2 * 512 * 256
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "FAIL: " + repr(e) + "\n" + traceback.format_exc()
    finally:
        dist.destroy_process_group()


def test_two_ranks_sharded_match_single_engine():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)
