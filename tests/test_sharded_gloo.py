"""The exchange of the catalog-sharded data-parallel step on CPU, world sizes 2 and 8 over gloo: the PRODUCT's collective schedule
(tcar_amd.sharded.ShardExchange.step — the same object ShardedEngine drives with the HIP entry points) with fp64 torch pieces,
against the single-process computation on the concatenated batch (no GPU, no HIP library: this pins the sequencing, the buffer
shapes of uneven / empty shards, the asynchronous item-row all-gather, and the ALGEBRA of the split —
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


def _worker(rank, world, port, ret, N=300, B=7):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tcar_amd  # noqa: F401
        from tcar_amd.dp import shard_bounds
        from tcar_amd.sharded import ShardExchange, shard_rows
        torch.manual_seed(3)
        ek, ldh, K, T = 24, 8, 5, 2                              # world 2: 7 sessions = 4 + 3, cap = 4 (one padding session)
        E = torch.randn(N, ek, dtype=torch.float64) * 0.3
        att = torch.tanh(torch.randn(B, ek, dtype=torch.float64))
        lab = torch.randint(0, N, (B,))
        neg = torch.randint(0, N, (B, K))
        coef = torch.randn(B, dtype=torch.float64)
        seq = torch.randint(1, N + 1, (B, T))                    # 1-based ids of the session-side gathers
        # ---- single-process reference on the whole batch
        Er, ar = E.clone().requires_grad_(True), att.clone().requires_grad_(True)
        ce_ref = torch.nn.functional.cross_entropy(ar @ Er.T, lab, reduction="none")
        ce_ref.sum().backward()
        dense_ref = Er.grad[:, :ldh].clone()
        for b in range(B):
            for k in range(K):
                dense_ref[neg[b, k]] += coef[b] * att[b, :ldh]       # densified negative-gather part (model_combine.py:142)
        sq_ref = float((dense_ref * dense_ref).sum())              # S5: the dense block's norm BEFORE any gathered row lands
        row_grad = lambda b, t: ar.grad[b, :ldh] * (t + 1)          # a stand-in for the session backward: any function of dX
        item_ref = dense_ref.clone()
        for b in range(B):
            for t in range(T):
                item_ref[seq[b, t] - 1] += row_grad(b, t)
        arena_ref = ar.grad.sum(0)                                  # a dense-weight gradient: the sum over ALL sessions
        # ---- the product's exchange schedule (ShardExchange.step) with fp64 torch pieces
        S = shard_rows(N, world)
        assert S == (256 if world == 2 else 5760)
        n0 = rank * S
        nl = min(N, n0 + S) - n0
        if world == 8:                                            # the Globo catalog at 8 ranks: seven 5,760-row shards + a SHORT last one
            assert nl == (5760 if rank < 7 else 46033 - 7 * 5760) and 0 < 46033 - 7 * 5760 < 5760
        lo, hi, cap = shard_bounds(B, world, rank)
        nloc = hi - lo
        ld_head = ek + 2 + K
        ldr = ldh + 1
        st = {}

        class Pieces:
            def begin(_):                                         # tcar_shard_begin: packed [attout | label | coef | negatives]
                head = torch.zeros(cap, ld_head, dtype=torch.float64)
                head[:, ek] = -1
                head[:, ek + 2:] = -1
                head[:nloc, :ek] = att[lo:hi]
                head[:nloc, ek] = lab[lo:hi].double()
                head[:nloc, ek + 1] = coef[lo:hi]
                head[:nloc, ek + 2:] = neg[lo:hi].double()
                return head

            def score(_, head_all):                               # tcar_shard_score: logits of the shard + its softmax statistics
                Bq = world * cap
                assert head_all.shape == (Bq, ld_head)
                st["att"], st["lab"] = head_all[:, :ek], head_all[:, ek].long()
                st["coef"], st["neg"] = head_all[:, ek + 1], head_all[:, ek + 2:].long()
                logits = st["att"] @ E[n0:n0 + nl].T
                m = logits.max(1).values
                ssum = torch.exp(logits - m[:, None]).sum(1)
                here = (st["lab"] >= n0) & (st["lab"] < n0 + nl)
                labl = torch.where(here, logits[torch.arange(Bq), (st["lab"] - n0).clamp(0, nl - 1)], torch.zeros(Bq, dtype=torch.float64))
                st["logits"], st["here"] = logits, here
                return torch.stack([m, ssum, labl], 1)

            def backward(_, stats_all):                           # tcar_softmax_combine + _grad, dE (local), dX partial
                assert stats_all.shape == (world, world * cap, 3)
                M = stats_all[:, :, 0].max(0).values
                lse = M + torch.log((stats_all[:, :, 1] * torch.exp(stats_all[:, :, 0] - M)).sum(0))
                st["ce"] = lse - stats_all[:, :, 2].sum(0)
                lse = torch.where(st["lab"] < 0, torch.full_like(lse, float("inf")), lse)   # padding sessions: zero gradient row
                dl = torch.exp(st["logits"] - lse[:, None])
                dl[st["here"], (st["lab"] - n0)[st["here"]]] -= 1.0
                assert float(dl[st["lab"] < 0].abs().max() if (st["lab"] < 0).any() else 0.0) == 0.0
                st["dE"] = dl.T @ st["att"]
                return dl @ E[n0:n0 + nl]

            def finish(_):                                        # tcar_shard_finish: negative rows of the shard, shard norm
                dense = st["dE"][:, :ldh].clone()
                for b in range(world * cap):
                    for k in range(K):
                        n = int(st["neg"][b, k]) - n0
                        if 0 <= n < nl and st["lab"][b] >= 0:
                            dense[n] += st["coef"][b] * st["att"][b, :ldh]
                st["dense"] = dense
                st["sq"] = (dense * dense).sum()                  # BEFORE the gathered rows are scattered in (S5)

            def session_backward(_, dx_rows):                     # tcar_step_session_backward: packed [row | id]
                assert dx_rows.shape == (cap, ek)
                st["dx"] = dx_rows
                rows = torch.zeros(cap * T, ldr, dtype=torch.float64)          # id 0 = padding
                for b in range(nloc):
                    for t in range(T):
                        rows[b * T + t, :ldh] = dx_rows[b, :ldh] * (t + 1)
                        rows[b * T + t, ldh] = float(seq[lo + b, t])
                st["arena"] = torch.cat([dx_rows[:nloc].sum(0), st["sq"].reshape(1)])   # arena gradient + the shard's norm piece
                return rows

            def scatter(_, all_rows):                             # tcar_scatter_add_rows_packed: owners keep theirs
                assert all_rows.shape == (world * cap * T, ldr)
                for r in all_rows:
                    n = int(r[ldh]) - 1 - n0
                    if int(r[ldh]) > 0 and 0 <= n < nl:
                        st["dense"][n] += r[:ldh]

            def arena(_):
                return st["arena"]

            def norms(_):
                pass

            def update(_):                                        # owners "update" their rows: x - 0.5 g
                stage = torch.zeros(world, S, ldh, dtype=torch.float64)
                stage[rank, :nl] = E[n0:n0 + nl, :ldh] - 0.5 * st["dense"]
                st["stage"] = None
                return stage, lambda full: st.__setitem__("stage", full.clone())

        xch = ShardExchange(dist.group.WORLD)
        assert xch.world == world and xch.rank == rank and not xch.use_reduce_scatter          # gloo: dX through an all-reduce
        xch.step(Pieces(), cap, update=True)
        assert xch.order == ["attout+labels+negatives", "softmax_stats", "dX", "rows+ids", "arena", "item_rows"]
        assert st["stage"] is None                                # collective 6 is asynchronous: installed by wait_rows
        xch.wait_rows()
        # ---- against the single-process computation
        assert torch.allclose(st["ce"][rank * cap:rank * cap + nloc], ce_ref[lo:hi].detach(), rtol=1e-12, atol=1e-12)
        assert torch.allclose(st["dx"][:nloc], ar.grad[lo:hi], rtol=1e-11, atol=1e-13)
        assert torch.allclose(st["dE"], Er.grad[n0:n0 + nl], rtol=1e-11, atol=1e-13)
        assert torch.allclose(st["dense"], item_ref[n0:n0 + nl], rtol=1e-11, atol=1e-13)
        assert torch.allclose(st["arena"][:ek], arena_ref, rtol=1e-11, atol=1e-13)
        assert abs(float(st["arena"][ek]) - sq_ref) <= 1e-10 * sq_ref            # sum over the shards of the dense norms
        full = st["stage"].view(-1, ldh)[:N]
        assert torch.allclose(full, E[:, :ldh] - 0.5 * item_ref, rtol=1e-11, atol=1e-13)
        assert xch.bytes_moved["item_rows"] == world * S * ldh * 8
        # EMPTY ranks: a batch of one session — every rank but 0 contributes padding only and still joins every collective
        B1 = 1
        lo, hi, cap = shard_bounds(B1, world, rank)
        nloc = hi - lo
        assert (rank == 0 and nloc == 1) or (rank > 0 and nloc == 0)
        xch.step(Pieces(), cap, update=False)
        assert xch.order == ["attout+labels+negatives", "softmax_stats", "dX", "rows+ids", "arena"]
        if rank == 0:
            one = torch.nn.functional.cross_entropy((att[:1] @ E.T), lab[:1], reduction="none")
            assert torch.allclose(st["ce"][:1], one, rtol=1e-12, atol=1e-12)
        # per-collective timing (bench.py's event-instrumented pass): off by default, one span per collective and step when on
        assert xch.collective_ms() == {}
        xch.timing = True
        lo, hi, cap = shard_bounds(B, world, rank)
        nloc = hi - lo
        xch.step(Pieces(), cap, update=True)
        xch.wait_rows()
        ms = xch.collective_ms()
        assert set(ms) == {"attout+labels+negatives", "softmax_stats", "dX", "rows+ids", "arena", "item_rows (issue)"}
        assert all(v >= 0.0 for v in ms.values())
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "FAIL: " + repr(e) + "\n" + traceback.format_exc()
    finally:
        dist.destroy_process_group()


def _run(world, N, B):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, N, B), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)


def test_sharded_exchange_algebra_two_ranks_gloo():
    _run(2, 300, 7)


def test_sharded_exchange_seams_eight_ranks_gloo():
    """The W = 8 seams of the benchmarked job: the Globo catalog cut into seven 5,760-row shards and a SHORT
    last shard (5,713 rows), 21 sessions over 8 ranks (cap 3: seven full ranks and an EMPTY eighth, three padding rows in the
    all-gathered batch), labels / negatives / gathered rows falling into every shard — the same ShardExchange.step the GPU engine
    drives, against the single-process computation; then a one-session batch (seven empty ranks)."""
    _run(8, 46033, 21)
