"""The exchange of the catalog-sharded data-parallel step (tcar_amd.sharded) on CPU, world size 2 over gloo, in fp64 against
the single-process computation on the concatenated batch (no GPU, no HIP library: this pins the ALGEBRA of the split —
per-shard softmax statistics and their combine, padding sessions of uneven shards, local dE, summed dX, the negative-term rows
that fall into a shard, and the S5 norm of the dense item block as a sum over the shards)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tcar_amd  # noqa: F401
        from tcar_amd.dp import shard_bounds
        from tcar_amd.sharded import shard_rows
        torch.manual_seed(3)
        N, ek, ldh, B, K = 300, 24, 8, 7, 5                     # 7 sessions over 2 ranks: 4 + 3, cap = 4 (one padding session)
        E = torch.randn(N, ek, dtype=torch.float64) * 0.3
        att = torch.tanh(torch.randn(B, ek, dtype=torch.float64))
        lab = torch.randint(0, N, (B,))
        neg = torch.randint(0, N, (B, K))
        coef = torch.randn(B, dtype=torch.float64)
        # ---- single-process reference on the whole batch
        Er, ar = E.clone().requires_grad_(True), att.clone().requires_grad_(True)
        ce_ref = torch.nn.functional.cross_entropy(ar @ Er.T, lab, reduction="none")
        ce_ref.sum().backward()
        dense_ref = Er.grad[:, :ldh].clone()
        for b in range(B):
            for k in range(K):
                dense_ref[neg[b, k]] += coef[b] * att[b, :ldh]       # densified negative-gather part (model_combine.py:142)
        # ---- the sharded exchange
        S = 128 * ((N + world - 1) // world // 128 + 1)           # shard_rows for this toy size
        assert shard_rows(N, world) == 256 and S == 256
        n0 = rank * S
        nl = min(N, n0 + S) - n0
        lo, hi, cap = shard_bounds(B, world, rank)
        att_loc = torch.zeros(cap, ek, dtype=torch.float64)
        att_loc[:hi - lo] = att[lo:hi]
        lab_loc = torch.full((cap,), -1, dtype=torch.int64)
        lab_loc[:hi - lo] = lab[lo:hi]
        neg_loc = torch.full((cap, K), -1, dtype=torch.int64)
        neg_loc[:hi - lo] = neg[lo:hi]
        coef_loc = torch.zeros(cap, dtype=torch.float64)
        coef_loc[:hi - lo] = coef[lo:hi]

        def gather(t):
            out = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(out, t)
            return torch.cat(out)

        att_all, lab_all, neg_all, coef_all = gather(att_loc), gather(lab_loc), gather(neg_loc), gather(coef_loc)
        Bq = world * cap
        logits = att_all @ E[n0:n0 + nl].T                                       # this shard's scores of EVERY session
        m = logits.max(1).values
        ssum = torch.exp(logits - m[:, None]).sum(1)
        here = (lab_all >= n0) & (lab_all < n0 + nl)
        labl = torch.where(here, logits[torch.arange(Bq), (lab_all - n0).clamp(0, nl - 1)], torch.zeros(Bq, dtype=torch.float64))
        stats_all = gather(torch.stack([m, ssum, labl], 1)).view(world, Bq, 3)   # tcar_softmax_stats + all-gather
        M = stats_all[:, :, 0].max(0).values                                      # tcar_softmax_combine
        lse = M + torch.log((stats_all[:, :, 1] * torch.exp(stats_all[:, :, 0] - M)).sum(0))
        ce = lse - stats_all[:, :, 2].sum(0)
        lse = torch.where(lab_all < 0, torch.full_like(lse, float("inf")), lse)   # padding sessions: zero gradient row
        mine = ce[rank * cap:rank * cap + (hi - lo)]
        assert torch.allclose(mine, ce_ref[lo:hi].detach(), rtol=1e-12, atol=1e-12)
        dl = torch.exp(logits - lse[:, None])                                     # tcar_softmax_grad
        dl[here, (lab_all - n0)[here]] -= 1.0
        assert float(dl[lab_all < 0].abs().max() if (lab_all < 0).any() else 0.0) == 0.0
        dE = dl.T @ att_all                                                       # stays local
        dX = dl @ E[n0:n0 + nl]                                                   # summed over the ranks
        dist.all_reduce(dX)
        assert torch.allclose(dX[rank * cap:rank * cap + (hi - lo)], ar.grad[lo:hi], rtol=1e-11, atol=1e-13)
        assert torch.allclose(dE, Er.grad[n0:n0 + nl], rtol=1e-11, atol=1e-13)
        dense = dE[:, :ldh].clone()                                               # tcar_neg_scatter_range
        for b in range(Bq):
            for k in range(K):
                n = int(neg_all[b, k]) - n0
                if 0 <= n < nl and lab_all[b] >= 0:
                    dense[n] += coef_all[b] * att_all[b, :ldh]
        assert torch.allclose(dense, dense_ref[n0:n0 + nl], rtol=1e-11, atol=1e-13)
        sq = (dense * dense).sum().reshape(1)                                     # S5: sum over the shards of the dense norms
        dist.all_reduce(sq)
        assert abs(float(sq) - float((dense_ref * dense_ref).sum())) <= 1e-10 * float(sq)
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "FAIL: " + repr(e) + "\n" + traceback.format_exc()
    finally:
        dist.destroy_process_group()


def test_sharded_exchange_algebra_two_ranks_gloo():
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)
