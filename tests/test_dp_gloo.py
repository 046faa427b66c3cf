"""World-size-2 test of the data-parallel gradient exchange (tcar_amd.dp.GradExchange) over gloo on CPU.

Each rank computes the gradients of ITS shard with the CPU oracle, the exchange combines them in the order
documented in dp.py, and the result must equal the single-process oracle on the concatenated global batch:
summed gradients, the IndexedSlices clip norms (S5) and the variables after clip + Adam."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

TIME = ["month_embedding", "day_embedding", "week_embedding", "hour_embedding", "minute_embedding"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_case():
    import tcar_amd  # noqa: F401
    from test_oracle_model import load_fixture
    z, params, batch = load_fixture()
    return z, params, batch


def _local_pieces(ora, sub):
    """Run the oracle on a shard and cut its gradients into the blocks the exchange moves."""
    from oracle.tcar_oracle import clip_rows
    out, grads, sqn = ora.loss_and_grads(sub)
    vals = out["_vals"]
    N, H, Ht = ora.N, ora.H, ora.Ht
    item_rows, item_dense = vals["item_emb"][0].grad, vals["item_emb"][1].grad       # [B,T,H], [N,H]
    d_et = out["cand_pt"].grad                                                         # [N, 5*Ht]
    big = torch.cat([item_dense.reshape(-1), d_et.reshape(-1)]).clone()
    arena, pieces = {}, {}
    for k, g in grads.items():
        if k == "item_emb":
            continue
        if k in TIME:                      # session-side (+ click-side) rows only; candidate rows come later
            i = TIME.index(k)
            tot = torch.zeros_like(g)
            sq = 0.0
            blocks = [(vals[k][0], torch.as_tensor(np.asarray(sub[["pm", "pd", "pw", "ph", "pmi"][i]]), dtype=torch.long))]
            if k == "week_embedding":
                blocks.append((vals[k][2], torch.as_tensor(np.asarray(sub["cw"]), dtype=torch.long)))
            if k == "hour_embedding":
                blocks.append((vals[k][2], torch.as_tensor(np.asarray(sub["ch"]), dtype=torch.long)))
            for rows, ids in blocks:
                tot.index_add_(0, ids.reshape(-1), rows.grad.reshape(-1, Ht))
                sq += float((rows.grad ** 2).sum())
            arena[k], pieces[k] = tot, sq
        else:
            arena[k] = g.clone()
            pieces[k] = sqn[k] if k in ("dec_pos", "duration_embedding") else 0.0
    pieces["item_emb"] = float((item_rows ** 2).sum())
    ids = torch.as_tensor(np.asarray(sub["seq"]), dtype=torch.int32).reshape(-1)
    return out, big, arena, pieces, ids, item_rows.reshape(-1, H).clone()


def _worker(rank, world, port, ret, split_big=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.tcar_oracle import TcarOracle, clip_rows
        from tcar_amd.dp import GradExchange, shard_bounds
        z, params, batch = _global_case()
        B = len(batch["label"])
        lo, hi, cap = shard_bounds(B, world, rank)
        sub = {k: v[lo:hi] for k, v in batch.items()}
        ora = TcarOracle(params, z["content"], z["mwdhm"], max_grad=1.5)
        N, H, Ht, T = ora.N, ora.H, ora.Ht, batch["seq"].shape[1]
        out, big, arena, pieces, ids, rows = _local_pieces(ora, sub)
        names = list(arena.keys())
        pnames = names + ["item_emb"]
        flat = torch.cat([arena[k].reshape(-1) for k in names] + [torch.tensor([pieces[k] for k in pnames], dtype=big.dtype)])
        # pad the sparse rows to the common capacity with id 0 (shards may be uneven)
        pad = cap * T - ids.numel()
        ids = torch.cat([ids, torch.zeros(pad, dtype=torch.int32)])
        rows = torch.cat([rows, torch.zeros(pad, H, dtype=rows.dtype)])
        state = {}

        def sqnorm_item():
            state["item_dense_sq"] = float((big[:N * H] ** 2).sum())

        def cand_time_bwd():
            d_et = big[N * H:].view(N, 5 * Ht)
            off = 0
            for k in names:
                n = arena[k].numel()
                if k in TIME:
                    i = TIME.index(k)
                    tab = ora.p[k].detach()
                    idx = ora.mwdhm[:, i]
                    r = tab[idx].clone().requires_grad_(True)
                    (gr,) = torch.autograd.grad(clip_rows(r), r, d_et[:, i * Ht:(i + 1) * Ht])
                    add = torch.zeros_like(tab).index_add_(0, idx, gr)
                    flat[off:off + n] += add.reshape(-1)
                    flat[len(flat) - len(pnames) + pnames.index(k)] += float((gr ** 2).sum())
                off += n

        def scatter_rows(all_ids, all_rows):
            ok = all_ids > 0
            big[:N * H].view(N, H).index_add_(0, (all_ids[ok] - 1).long(), all_rows[ok])

        # split_big: step 1 as two all-reduces, the candidate-time block first (the order DPEngine uses on the GPU)
        bigs = [big[N * H:], big[:N * H]] if split_big else big
        GradExchange(None).run(bigs, flat, ids, rows, sqnorm_item, cand_time_bwd, scatter_rows, lambda: None)
        # unpack the exchanged result into per-variable gradients and clip norms
        grads, sqn, off = {}, {}, 0
        for k in names:
            n = arena[k].numel()
            grads[k] = flat[off:off + n].view_as(arena[k]).clone()
            off += n
        pv = flat[off:]
        item = torch.zeros(N + 1, H, dtype=big.dtype)
        item[1:] = big[:N * H].view(N, H)
        grads["item_emb"] = item
        for i, k in enumerate(pnames):
            if k == "item_emb":
                sqn[k] = state["item_dense_sq"] + float(pv[i])
            elif k in TIME or k in ("dec_pos", "duration_embedding"):
                sqn[k] = float(pv[i])
            else:
                sqn[k] = float((grads[k] ** 2).sum())
        # reference: the single-process oracle on the whole batch
        ref = TcarOracle(params, z["content"], z["mwdhm"], max_grad=1.5)
        _, g_ref, sq_ref = ref.loss_and_grads(batch)
        for k in g_ref:
            np.testing.assert_allclose(grads[k].numpy(), g_ref[k].numpy(), rtol=1e-9, atol=1e-13, err_msg=k)
            np.testing.assert_allclose(sqn[k], sq_ref[k], rtol=1e-9, err_msg="sqn " + k)
        ora.apply_adam({k: grads[k] for k in ora.p}, sqn)
        ref.apply_adam(g_ref, sq_ref)
        for k, v in ref.export().items():
            np.testing.assert_allclose(ora.export()[k], v, rtol=1e-9, atol=1e-14, err_msg="adam " + k)
        ret[rank] = "ok"
    except Exception as e:                                  # surface the failure in the parent
        import traceback
        ret[rank] = "FAIL: " + repr(e) + "\n" + traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("split_big", [False, True])
def test_gradient_exchange_world2_matches_single_process(split_big):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, split_big), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)


def test_shard_bounds():
    from tcar_amd.dp import shard_bounds
    assert [shard_bounds(5, 2, r) for r in range(2)] == [(0, 3, 3), (3, 5, 3)]
    assert [shard_bounds(2, 4, r) for r in range(4)] == [(0, 1, 1), (1, 2, 1), (2, 2, 1), (2, 2, 1)]
    assert shard_bounds(512, 8, 7) == (448, 512, 64)
