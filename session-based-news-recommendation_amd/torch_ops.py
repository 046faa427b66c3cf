"""`torch.ops.tcar.*` — the op-level boundary of SURVEY.md §8(b) as registered PyTorch custom ops (torch.library), each
backed by the C-ABI launchers of include/tcar_hip.h, with autograd formulas registered through `register_autograd` (the
backward passes are custom ops themselves, so the graph stays composable) and shape functions (`register_fake`).

    torch.ops.tcar.gather_clip       modules.py:13-41 x 10 + the concats of model_combine.py:65,84,94,111
    torch.ops.tcar.attn_pool         modules.py:72-152, util.py:92-100 (single + multi attention pools)
    torch.ops.tcar.score_ce          model_combine.py:132-138,145: full-catalog logits + sparse softmax CE (train)
    torch.ops.tcar.score_rank        model_combine.py:283 + util.py:8-18 + :301: logits, rank, top-k, CE (eval)
    torch.ops.tcar.neg_term          model_combine.py:142-143: negative-feedback term
    torch.ops.tcar.clip_adam_        model_combine.py:155-163 for one variable: clip_by_norm + TF-1 Adam, in place
    torch.ops.tcar.rank_topk         util.py:13-17, model_combine.py:301
    torch.ops.tcar.linear            modules.py:43-70 (linear_2d / flattened linear_3d)

Importing this module registers the ops (idempotent).  Everything runs on the CURRENT torch stream of the tensors' device;
tensors are fp32 / int32, contiguous, on the GPU — there is no CPU implementation (`_lib.load` raises without the library,
and the CUDA dispatch key is the only kernel registered).  Layout conventions are those of include/tcar_hip.h (padded-concat
space: ldh = H rounded up to 64, ek = 2*ldh + 5*ldt).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import Batch, Dims, Grads, Segments, Tables, check

_ROWOFF = (0, 13, 45, 53, 78, 139)          # month, day, week, hour, minute, dwell rows inside the [150, ldt] block
_LIB = None


def _lib_():
    global _LIB
    if _LIB is None:
        _LIB = _lib.load()
    return _LIB


def _p(t: Tensor, off_bytes: int = 0):
    return C.c_void_p(t.data_ptr() + off_bytes)


def _st(t: Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _chk(*ts: Tensor):
    for t in ts:
        if not t.is_cuda or not t.is_contiguous():
            raise _lib.TcarError("tcar ops take contiguous GPU tensors (no CPU fallback)")


def _geom(E: Tensor, pos: Tensor, small: Tensor, H: int, Ht: int):
    ldh, ldt = pos.shape[1], small.shape[1]
    ek = E.shape[1]
    if ek != 2 * ldh + 5 * ldt or small.shape[0] != 150 or pos.shape[0] != 40:
        raise ValueError("tables must be E [N, 2*ldh + 5*ldt], pos [40, ldh], small [150, ldt]")
    return Dims(E.shape[0], H, Ht, ldh, ldt), ldh, ldt, ek


def _tables(E: Tensor, pos: Tensor, small: Tensor, ldt: int) -> Tables:
    t = Tables()
    t.E, t.pos = E.data_ptr(), pos.data_ptr()
    for k in range(5):
        t.time[k] = small.data_ptr() + 4 * _ROWOFF[k] * ldt
    t.dur = small.data_ptr() + 4 * _ROWOFF[5] * ldt
    return t


def _batch(feed: Tensor, B: int, T: int, K: int = 0, neg: Tensor = None, label: Tensor = None) -> Batch:
    """feed = int32 [7*B*T + 2*B]: seq | month | day | week | hour+1 | minute+1 | dwell bucket | click week | click hour
    (the feed_dict of model_combine.py:214-227, packed like TcarEngine.upload)."""
    bt = Batch()
    bt.B, bt.T, bt.K = B, T, K
    base, n = feed.data_ptr(), B * T
    bt.seq = base
    for k in range(5):
        bt.pub[k] = base + 4 * (k + 1) * n
    bt.gap = base + 4 * 6 * n
    bt.cw = base + 4 * 7 * n
    bt.ch = base + 4 * (7 * n + B)
    bt.label = label.data_ptr() if label is not None else None
    bt.neg = neg.data_ptr() if neg is not None else None
    return bt


# ----------------------------------------------------------------------------------------------- gather_clip
@torch.library.custom_op("tcar::gather_clip", mutates_args=(), device_types="cuda")
def gather_clip(E: Tensor, pos: Tensor, small: Tensor, feed: Tensor, B: int, T: int, H: int, Ht: int) -> List[Tensor]:
    """-> [x_icp [B*T, 2*ldh] = clip(item)+clip(pos) | clip(content),  x_pt [B*T, 5*ldt],  x_act [B*T, ldt],
    click_t [B, 2*ldt]]; every lookup is tf.nn.embedding_lookup(..., max_norm=1) (modules.py:36)."""
    _chk(E, pos, small, feed)
    dims, ldh, ldt, ek = _geom(E, pos, small, H, Ht)
    f32 = dict(dtype=torch.float32, device=E.device)
    x_icp, x_pt = torch.empty(B * T, 2 * ldh, **f32), torch.empty(B * T, 5 * ldt, **f32)
    x_act, click = torch.empty(B * T, ldt, **f32), torch.empty(B, 2 * ldt, **f32)
    tab, bt = _tables(E, pos, small, ldt), _batch(feed, B, T)
    check(_lib_().tcar_gather_clip_fwd(C.byref(dims), C.byref(tab), C.byref(bt), _p(x_icp), _p(x_pt), _p(x_act), _p(click),
                                       _st(E)), "tcar_gather_clip_fwd")
    return [x_icp, x_pt, x_act, click]


@gather_clip.register_fake
def _(E, pos, small, feed, B, T, H, Ht):
    ldh, ldt = pos.shape[1], small.shape[1]
    return [E.new_empty(B * T, 2 * ldh), E.new_empty(B * T, 5 * ldt), E.new_empty(B * T, ldt), E.new_empty(B, 2 * ldt)]


@torch.library.custom_op("tcar::gather_clip_bwd", mutates_args=(), device_types="cuda")
def gather_clip_bwd(E: Tensor, pos: Tensor, small: Tensor, feed: Tensor, dx_icp: Tensor, dx_pt: Tensor, dx_act: Tensor,
                    dclick: Tensor, B: int, T: int, H: int, Ht: int) -> List[Tensor]:
    """IndexedSlices gradients of the lookups, densified: [g_item [N, ldh] (row n = item id n+1), g_pos [40, ldh],
    g_small [150, ldt], sqn [32] = per-variable sum of ||row gradient||^2 (slot 0 item, 1 pos, 2..6 time tables, 7 dwell:
    the norm tf.clip_by_norm sees for an IndexedSlices)]."""
    _chk(E, pos, small, feed, dx_icp, dx_pt, dx_act, dclick)
    dims, ldh, ldt, ek = _geom(E, pos, small, H, Ht)
    f32 = dict(dtype=torch.float32, device=E.device)
    g_item, g_pos = torch.zeros(E.shape[0], ldh, **f32), torch.zeros(40, ldh, **f32)
    g_small, sqn = torch.zeros(150, ldt, **f32), torch.zeros(_lib.NSLOT, **f32)
    gr = Grads()
    gr.g_item, gr.g_pos, gr.sqn = g_item.data_ptr(), g_pos.data_ptr(), sqn.data_ptr()
    for k in range(5):
        gr.g_time[k] = g_small.data_ptr() + 4 * _ROWOFF[k] * ldt
        gr.slot_time[k] = 2 + k
    gr.g_dur = g_small.data_ptr() + 4 * _ROWOFF[5] * ldt
    gr.slot_item, gr.slot_pos, gr.slot_dur = 0, 1, 7
    tab, bt = _tables(E, pos, small, ldt), _batch(feed, B, T)
    check(_lib_().tcar_gather_clip_bwd(C.byref(dims), C.byref(tab), C.byref(bt), _p(dx_icp), _p(dx_pt), _p(dx_act),
                                       _p(dclick), C.byref(gr), _st(E)), "tcar_gather_clip_bwd")
    return [g_item, g_pos, g_small, sqn]


@gather_clip_bwd.register_fake
def _(E, pos, small, feed, dx_icp, dx_pt, dx_act, dclick, B, T, H, Ht):
    return [E.new_empty(E.shape[0], pos.shape[1]), torch.empty_like(pos), torch.empty_like(small), E.new_empty(_lib.NSLOT)]


def _gather_setup(ctx, inputs, output):
    E, pos, small, feed, B, T, H, Ht = inputs
    ctx.save_for_backward(E, pos, small, feed)
    ctx.args = (B, T, H, Ht)


def _gather_backward(ctx, grads):
    E, pos, small, feed = ctx.saved_tensors
    B, T, H, Ht = ctx.args
    ldh, ldt = pos.shape[1], small.shape[1]
    z = lambda g, shape: g.contiguous() if g is not None else torch.zeros(shape, dtype=torch.float32, device=E.device)
    dx_icp, dx_pt = z(grads[0], (B * T, 2 * ldh)), z(grads[1], (B * T, 5 * ldt))
    dx_act, dclick = z(grads[2], (B * T, ldt)), z(grads[3], (B, 2 * ldt))
    g_item, g_pos, g_small, _ = torch.ops.tcar.gather_clip_bwd(E, pos, small, feed, dx_icp, dx_pt, dx_act, dclick, B, T, H, Ht)
    dE = torch.zeros_like(E)
    dE[:, :ldh] = g_item                       # the content and candidate-time columns of E are not trained through lookups
    return dE, g_pos, g_small, None, None, None, None, None


gather_clip.register_autograd(_gather_backward, setup_context=_gather_setup)


# ------------------------------------------------------------------------------------------------- attn_pool
@torch.library.custom_op("tcar::attn_pool", mutates_args=(), device_types="cuda")
def attn_pool(x_icp: Tensor, x_pt: Tensor, pre1: Tensor, pre2: Tensor, q: Tensor, w_res1: Tensor, w_res2: Tensor,
              H: int) -> Tuple[Tensor, Tensor]:
    """x_icp [B,T,2*ldh], x_pt [B,T,5*ldt], pre1 / pre2 [B,T,ldh], q [B,2*ldh], w_res* [ldh] -> (pooled [B, ek], alpha [3,B*T])"""
    _chk(x_icp, x_pt, pre1, pre2, q, w_res1, w_res2)
    B, T, ic = x_icp.shape
    ldt = x_pt.shape[2] // 5
    dims = Dims(1, H, ldt, ic // 2, ldt)
    pooled = torch.empty(B, ic + 5 * ldt, dtype=torch.float32, device=x_icp.device)
    alpha = torch.empty(3, B * T, dtype=torch.float32, device=x_icp.device)
    check(_lib_().tcar_attn_pool_fwd(C.byref(dims), B, T, _p(x_icp), _p(x_pt), _p(pre1), _p(pre2), _p(q), _p(w_res1),
                                     _p(w_res2), _p(pooled), _p(alpha), _st(x_icp)), "tcar_attn_pool_fwd")
    return pooled, alpha


@attn_pool.register_fake
def _(x_icp, x_pt, pre1, pre2, q, w_res1, w_res2, H):
    B, T, ic = x_icp.shape
    return x_icp.new_empty(B, ic + x_pt.shape[2]), x_icp.new_empty(3, B * T)


@torch.library.custom_op("tcar::attn_pool_bwd", mutates_args=(), device_types="cuda")
def attn_pool_bwd(x_icp: Tensor, x_pt: Tensor, pre1: Tensor, pre2: Tensor, q: Tensor, w_res1: Tensor, w_res2: Tensor,
                  alpha: Tensor, dpooled: Tensor, H: int) -> List[Tensor]:
    """-> [dx_icp, dx_pt, dpre1, dpre2, dq, g_wres1, g_wres2]"""
    _chk(x_icp, x_pt, pre1, pre2, q, w_res1, w_res2, alpha, dpooled)
    B, T, ic = x_icp.shape
    ldt = x_pt.shape[2] // 5
    dims = Dims(1, H, ldt, ic // 2, ldt)
    dx_icp, dx_pt = torch.empty_like(x_icp), torch.empty_like(x_pt)
    dq, dpre1, dpre2 = torch.empty_like(q), torch.empty_like(pre1), torch.empty_like(pre2)
    g1, g2 = torch.zeros_like(w_res1), torch.zeros_like(w_res2)
    check(_lib_().tcar_attn_pool_bwd(C.byref(dims), B, T, _p(x_icp), _p(x_pt), _p(pre1), _p(pre2), _p(q), _p(w_res1),
                                     _p(w_res2), _p(alpha), _p(dpooled), _p(dx_icp), _p(dx_pt), _p(dq), _p(dpre1), _p(dpre2),
                                     _p(g1), _p(g2), _st(x_icp)), "tcar_attn_pool_bwd")
    return [dx_icp, dx_pt, dpre1, dpre2, dq, g1, g2]


@attn_pool_bwd.register_fake
def _(x_icp, x_pt, pre1, pre2, q, w_res1, w_res2, alpha, dpooled, H):
    return [torch.empty_like(t) for t in (x_icp, x_pt, pre1, pre2, q, w_res1, w_res2)]


def _pool_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs[:7], output[1])
    ctx.H = inputs[7]


def _pool_backward(ctx, dpooled, dalpha):
    x_icp, x_pt, pre1, pre2, q, w1, w2, alpha = ctx.saved_tensors
    out = torch.ops.tcar.attn_pool_bwd(x_icp, x_pt, pre1, pre2, q, w1, w2, alpha, dpooled.contiguous(), ctx.H)
    return out[0], out[1], out[2], out[3], out[4], out[5], out[6], None


attn_pool.register_autograd(_pool_backward, setup_context=_pool_setup)


# -------------------------------------------------------------------------------------------------- score_ce
def _logits(attout: Tensor, E: Tensor, n_items: int) -> Tensor:
    B, ek = attout.shape
    npad = (n_items + 3) // 4 * 4
    logits = torch.empty(B, npad, dtype=torch.float32, device=attout.device)
    check(_lib_().tcar_gemm_f32(1, B, n_items, ek, _p(attout), ek, _p(E), E.shape[1], _p(logits), npad, None, 0, 0, 1,
                                _st(attout)), "tcar_gemm_f32")
    return logits


@torch.library.custom_op("tcar::score_ce", mutates_args=(), device_types="cuda")
def score_ce(attout: Tensor, E: Tensor, label: Tensor) -> Tuple[Tensor, Tensor]:
    """Training form: logits = attout E^T over the whole catalog (model_combine.py:138), sparse softmax cross entropy
    (:145).  -> (ce [B], dlogits [B, ceil4(N)] = softmax - onehot, saved for the backward pass).  fp32 MFMA."""
    _chk(attout, E, label)
    N = E.shape[0]
    logits = _logits(attout, E, N)
    ce = torch.empty(attout.shape[0], dtype=torch.float32, device=attout.device)
    check(_lib_().tcar_softmax_ce(attout.shape[0], N, _p(logits), logits.shape[1], _p(label), _p(ce), _st(attout)),
          "tcar_softmax_ce")
    return ce, logits


@score_ce.register_fake
def _(attout, E, label):
    return attout.new_empty(attout.shape[0]), attout.new_empty(attout.shape[0], (E.shape[0] + 3) // 4 * 4)


@torch.library.custom_op("tcar::score_ce_bwd", mutates_args=(), device_types="cuda")
def score_ce_bwd(dlogits: Tensor, attout: Tensor, E: Tensor) -> Tuple[Tensor, Tensor]:
    """dattout = dlogits E, dE = dlogits^T attout (both contractions on the fp32 MFMA GEMM)"""
    _chk(dlogits, attout, E)
    B, ek = attout.shape
    N, npad = E.shape[0], dlogits.shape[1]
    dattout, dE = torch.empty_like(attout), torch.empty_like(E)
    st = _st(attout)
    # the catalog-long contraction runs over ceil4(N) (the k-contiguous operand needs K % 4 == 0): the pad columns of dlogits
    # are zero (tcar_softmax_ce) and E is given zero rows to match
    Ek = E if npad == N else torch.cat([E, E.new_zeros(npad - N, ek)])
    check(_lib_().tcar_gemm_f32(0, B, ek, npad, _p(dlogits), npad, _p(Ek), ek, _p(dattout), ek, None, 0, 0, 1, st), "tcar_gemm_f32")
    check(_lib_().tcar_gemm_f32(2, N, ek, B, _p(dlogits), npad, _p(attout), ek, _p(dE), ek, None, 0, 0, 1, st), "tcar_gemm_f32")
    return dattout, dE


@score_ce_bwd.register_fake
def _(dlogits, attout, E):
    return torch.empty_like(attout), torch.empty_like(E)


def _ce_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], output[1])


def _ce_backward(ctx, dce, _dl):
    attout, E, dlogits = ctx.saved_tensors
    dattout, dE = torch.ops.tcar.score_ce_bwd((dlogits * dce[:, None]).contiguous(), attout, E)
    return dattout, dE, None


score_ce.register_autograd(_ce_backward, setup_context=_ce_setup)


@torch.library.custom_op("tcar::score_rank", mutates_args=(), device_types="cuda")
def score_rank(attout: Tensor, E: Tensor, label: Tensor, k: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Evaluation form (model_combine.py:283 + util.py:8-18 + :301): -> (rank [B] int32, topk [B,k] int32, ce [B]); the
    score matrix never leaves the device."""
    _chk(attout, E, label)
    B, N = attout.shape[0], E.shape[0]
    logits = _logits(attout, E, N)
    rank = torch.empty(B, dtype=torch.int32, device=attout.device)
    topk = torch.empty(B, k, dtype=torch.int32, device=attout.device)
    ce = torch.empty(B, dtype=torch.float32, device=attout.device)
    check(_lib_().tcar_eval_rows(B, N, _p(logits), logits.shape[1], _p(label), k, _p(rank), _p(topk), _p(ce), _st(attout)),
          "tcar_eval_rows")
    return rank, topk, ce


@score_rank.register_fake
def _(attout, E, label, k):
    B = attout.shape[0]
    return (attout.new_empty(B, dtype=torch.int32), attout.new_empty(B, k, dtype=torch.int32), attout.new_empty(B))


# -------------------------------------------------------------------------------------------------- neg_term
@torch.library.custom_op("tcar::neg_term", mutates_args=(), device_types="cuda")
def neg_term(E: Tensor, neg: Tensor, attout: Tensor, H: int, Ht: int) -> Tuple[Tensor, Tensor, Tensor]:
    """neg_fb[b] = -log(1 - sigmoid(x_b) + 1e-24), x_b = sum_k E[neg[b,k], 0:ic] . attout[b, 0:ic] (model_combine.py:142-143).
    -> (neg_fb [B], coef [B] = d neg_fb / d x_b, negpart [B, ic] = coef * sum_k E[neg[b,k], 0:ic])"""
    _chk(E, neg, attout)
    B, K = neg.shape
    ldt = 64 if Ht <= 64 else (128 if Ht <= 128 else 256)
    ldh = (E.shape[1] - 5 * ldt) // 2
    dims = Dims(E.shape[0], H, Ht, ldh, ldt)
    f32 = dict(dtype=torch.float32, device=E.device)
    fb, coef, part = torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, 2 * ldh, **f32)
    check(_lib_().tcar_neg_fwd(C.byref(dims), B, K, _p(E), _p(neg), _p(attout), 1.0, _p(fb), _p(coef), _p(part), _st(E)),
          "tcar_neg_fwd")
    return fb, coef, part


@neg_term.register_fake
def _(E, neg, attout, H, Ht):
    B = neg.shape[0]
    ldt = 64 if Ht <= 64 else (128 if Ht <= 128 else 256)
    return E.new_empty(B), E.new_empty(B), E.new_empty(B, E.shape[1] - 5 * ldt)


@torch.library.custom_op("tcar::neg_term_bwd", mutates_args=(), device_types="cuda")
def neg_term_bwd(E: Tensor, neg: Tensor, attout: Tensor, coef: Tensor, H: int, Ht: int) -> Tensor:
    """g_item [N, ldh] += coef[b] * attout[b, 0:ldh] at rows neg[b, k] (the item columns; content is frozen)"""
    _chk(E, neg, attout, coef)
    B, K = neg.shape
    ldt = 64 if Ht <= 64 else (128 if Ht <= 128 else 256)
    ldh = (E.shape[1] - 5 * ldt) // 2
    dims = Dims(E.shape[0], H, Ht, ldh, ldt)
    g_item = torch.zeros(E.shape[0], ldh, dtype=torch.float32, device=E.device)
    check(_lib_().tcar_neg_scatter(C.byref(dims), B, K, _p(neg), _p(attout), _p(coef), _p(g_item), None, None, 1.0, None,
                                   _st(E)), "tcar_neg_scatter")
    return g_item


@neg_term_bwd.register_fake
def _(E, neg, attout, coef, H, Ht):
    ldt = 64 if Ht <= 64 else (128 if Ht <= 128 else 256)
    return E.new_empty(E.shape[0], (E.shape[1] - 5 * ldt) // 2)


def _neg_setup(ctx, inputs, output):
    E, neg, attout, H, Ht = inputs
    ctx.save_for_backward(E, neg, attout, output[1], output[2])
    ctx.hh = (H, Ht)


def _neg_backward(ctx, dfb, _dc, _dp):
    E, neg, attout, coef, part = ctx.saved_tensors
    H, Ht = ctx.hh
    c = (coef * dfb).contiguous()
    g_item = torch.ops.tcar.neg_term_bwd(E, neg, attout, c, H, Ht)
    dE = torch.zeros_like(E)
    dE[:, :g_item.shape[1]] = g_item
    dattout = torch.zeros_like(attout)
    dattout[:, :part.shape[1]] = part * dfb[:, None]
    return dE, None, dattout, None, None


neg_term.register_autograd(_neg_backward, setup_context=_neg_setup)


# ------------------------------------------------------------------------------------------------- clip_adam_
@torch.library.custom_op("tcar::clip_adam_", mutates_args=("w", "m", "v"), device_types="cuda")
def clip_adam_(w: Tensor, g: Tensor, m: Tensor, v: Tensor, sqnorm_pieces: float, use_dense_norm: bool, clip: float,
               lr_t: float, b1: float, b2: float, eps: float) -> None:
    """One variable of model_combine.py:155-163: g <- tf.clip_by_norm(g, clip) with norm^2 = (use_dense_norm ? ||g||^2 : 0)
    + sqnorm_pieces (the IndexedSlices pieces of a table), then TF-1 Adam with the bias-corrected rate lr_t;
    w, m, v are updated in place (flat fp32, numel % 4 == 0)."""
    _chk(w, g, m, v)
    n = w.numel()
    if n % 4:
        raise ValueError("clip_adam_: numel must be a multiple of 4 (pad the variable)")
    lib, st = _lib_(), _st(w)
    segs = Segments()
    segs.nseg = 1
    segs.off[0], segs.len[0], segs.slot[0] = 0, n, 0
    sq_dense = torch.zeros(_lib.NSLOT, dtype=torch.float32, device=w.device)
    pieces = torch.zeros(_lib.NSLOT, dtype=torch.float32, device=w.device)
    pieces[0] = sqnorm_pieces
    use = torch.zeros(_lib.NSLOT, dtype=torch.int32, device=w.device)
    use[0] = int(use_dense_norm)
    if use_dense_norm:
        check(lib.tcar_sqnorm(_p(g), C.byref(segs), _p(sq_dense), st), "tcar_sqnorm")
    check(lib.tcar_clip_adam(_p(w), _p(g), _p(m), _p(v), C.byref(segs), _p(sq_dense), _p(pieces), _p(use), clip, lr_t, b1, b2,
                             eps, st), "tcar_clip_adam")


# ------------------------------------------------------------------------------------------ rank_topk, linear
@torch.library.custom_op("tcar::rank_topk", mutates_args=(), device_types="cuda")
def rank_topk(logits: Tensor, label: Tensor, n_valid: int, k: int) -> Tuple[Tensor, Tensor]:
    """rank[b] = 1 + #{n < n_valid: logits[b,n] > logits[b,label[b]]}; topk in np.argsort(x)[::-1] order (ties: higher index first)"""
    _chk(logits, label)
    B, ld = logits.shape
    rank = torch.empty(B, dtype=torch.int32, device=logits.device)
    topk = torch.empty(B, k, dtype=torch.int32, device=logits.device)
    check(_lib_().tcar_rank_topk(B, n_valid, _p(logits), ld, _p(label), k, _p(rank), _p(topk), _st(logits)), "tcar_rank_topk")
    return rank, topk


@rank_topk.register_fake
def _(logits, label, n_valid, k):
    B = logits.shape[0]
    return logits.new_empty(B, dtype=torch.int32), logits.new_empty(B, k, dtype=torch.int32)


@torch.library.custom_op("tcar::linear", mutates_args=(), device_types="cuda")
def linear(x: Tensor, w: Tensor, bias: Tensor, act: int) -> Tensor:
    """act(x @ w + bias), x [M,K], w [K,N], bias [N] (pass zeros for linear_3d, which has none); act 0 none, 1 relu, 2 tanh"""
    _chk(x, w, bias)
    M, K = x.shape
    N = w.shape[1]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    check(_lib_().tcar_gemm_f32(0, M, N, K, _p(x), K, _p(w), N, _p(y), N, _p(bias), act, 0, 1, _st(x)), "tcar_gemm_f32")
    return y


@linear.register_fake
def _(x, w, bias, act):
    return x.new_empty(x.shape[0], w.shape[1])


@torch.library.custom_op("tcar::linear_bwd", mutates_args=(), device_types="cuda")
def linear_bwd(dy: Tensor, x: Tensor, w: Tensor, y: Tensor, act: int) -> Tuple[Tensor, Tensor, Tensor]:
    _chk(dy, x, w, y)
    M, K = x.shape
    N = w.shape[1]
    lib, st = _lib_(), _st(x)
    dz, db = dy.clone(), torch.zeros(N, dtype=torch.float32, device=x.device)
    if act:
        check(lib.tcar_dact_colsum(M, N, N, _p(y), _p(dz), _p(db), act, st), "tcar_dact_colsum")
    else:
        db = dz.sum(0)
    dx, dw = torch.empty_like(x), torch.empty_like(w)
    check(lib.tcar_gemm_f32(1, M, K, N, _p(dz), N, _p(w), N, _p(dx), K, None, 0, 0, 1, st), "tcar_gemm_f32")
    check(lib.tcar_gemm_f32(2, K, N, M, _p(x), K, _p(dz), N, _p(dw), N, None, 0, 0, 1, st), "tcar_gemm_f32")
    return dx, dw, db


@linear_bwd.register_fake
def _(dy, x, w, y, act):
    return torch.empty_like(x), torch.empty_like(w), x.new_empty(w.shape[1])


def _lin_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], output)
    ctx.act = inputs[3]


def _lin_backward(ctx, dy):
    x, w, y = ctx.saved_tensors
    dx, dw, db = torch.ops.tcar.linear_bwd(dy.contiguous(), x, w, y, ctx.act)
    return dx, dw, db, None


linear.register_autograd(_lin_backward, setup_context=_lin_setup)

# --------------------------------------------------------- optional blocks of modules.py:194-336 (not on TCAR's graph)
@torch.library.custom_op("tcar::linear_residual", mutates_args=(), device_types="cuda")
def linear_residual(x: Tensor, w: Tensor, bias: Tensor, res: Tensor) -> Tensor:
    """x @ w + bias + res: the readout layer + residual of `feedforward` (modules.py:327-333) in one GEMM epilogue"""
    _chk(x, w, bias, res)
    M, K = x.shape
    N = w.shape[1]
    y = res.clone()                                   # the epilogue accumulates into it (beta = 1)
    check(_lib_().tcar_gemm_f32(0, M, N, K, _p(x), K, _p(w), N, _p(y), N, _p(bias), 0, 1, 1, _st(x)), "tcar_gemm_f32")
    return y


@linear_residual.register_fake
def _(x, w, bias, res):
    return torch.empty_like(res)


def _linres_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])


def _linres_backward(ctx, dy):
    x, w = ctx.saved_tensors
    dy = dy.contiguous()
    dx, dw, db = torch.ops.tcar.linear_bwd(dy, x, w, dy, 0)
    return dx, dw, db, dy


linear_residual.register_autograd(_linres_backward, setup_context=_linres_setup)


def feedforward(inputs: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """modules.py:306-336 with dropout off: two kernel-size-1 convolutions = per-position linear layers, relu inside,
    residual outside: relu(x w1 + b1) w2 + b2 + x.  inputs [N, T, C], w1 [C, F], w2 [F, C]."""
    N, T, Cc = inputs.shape
    x = inputs.reshape(N * T, Cc).contiguous()
    h = torch.ops.tcar.linear(x, w1, b1, 1)
    return torch.ops.tcar.linear_residual(h, w2, b2, x).reshape(N, T, Cc)


@torch.library.custom_op("tcar::normalize", mutates_args=(), device_types="cuda")
def normalize(x: Tensor, gamma: Tensor, beta: Tensor, epsilon: float) -> Tuple[Tensor, Tensor]:
    """modules.py:194-218: layer normalisation over the last axis -> (y, stats [M, 2] = (mean, 1/std) for the backward pass)"""
    _chk(x, gamma, beta)
    Cc = x.shape[-1]
    M = x.numel() // Cc
    y = torch.empty_like(x)
    stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
    check(_lib_().tcar_layernorm_fwd(M, Cc, _p(x), _p(gamma), _p(beta), epsilon, _p(y), _p(stats), _st(x)), "tcar_layernorm_fwd")
    return y, stats


@normalize.register_fake
def _(x, gamma, beta, epsilon):
    return torch.empty_like(x), x.new_empty(x.numel() // x.shape[-1], 2)


@torch.library.custom_op("tcar::normalize_bwd", mutates_args=(), device_types="cuda")
def normalize_bwd(x: Tensor, gamma: Tensor, stats: Tensor, dy: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    _chk(x, gamma, stats, dy)
    Cc = x.shape[-1]
    M = x.numel() // Cc
    dx, dg, db = torch.empty_like(x), torch.zeros_like(gamma), torch.zeros_like(gamma)
    check(_lib_().tcar_layernorm_bwd(M, Cc, _p(x), _p(gamma), _p(stats), _p(dy), _p(dx), _p(dg), _p(db), _st(x)),
          "tcar_layernorm_bwd")
    return dx, dg, db


@normalize_bwd.register_fake
def _(x, gamma, stats, dy):
    return torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)


def _ln_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], output[1])


def _ln_backward(ctx, dy, _ds):
    x, gamma, stats = ctx.saved_tensors
    dx, dg, db = torch.ops.tcar.normalize_bwd(x, gamma, stats, dy.contiguous())
    return dx, dg, db, None


normalize.register_autograd(_ln_backward, setup_context=_ln_setup)


# optional _lib.Tuning copy the optional ops below pass to the library (tests pin the scalar / MFMA form of mha_core with it);
# None = the process-wide switch values.  Python-side state of this module: the library itself keeps no mutable switch.
TUNING = None


def _tune_ptr():
    import ctypes as _C
    return _C.cast(_C.pointer(TUNING), _C.c_void_p) if TUNING is not None else None


@torch.library.custom_op("tcar::mha_core", mutates_args=(), device_types="cuda")
def mha_core(Q: Tensor, K: Tensor, V: Tensor, key_mask: Tensor, query_mask: Tensor, heads: int, causal: bool) -> Tuple[Tensor, Tensor]:
    """the attention core of modules.py:256-292 (scores, key / causal masks, softmax, query mask, weighted sum) on the
    matrix cores for head sizes 32 / 64 -> (O [N,Tq,C], P [N*heads,Tq,Tk])"""
    _chk(Q, K, V, key_mask, query_mask)
    N, Tq, Cc = Q.shape
    Tk = K.shape[1]
    O = torch.empty_like(Q)
    P = torch.empty(N * heads, Tq, Tk, dtype=torch.float32, device=Q.device)
    check(_lib_().tcar_mha_core_fwd_tuned(_tune_ptr(), N, Tq, Tk, Cc, heads, int(causal), _p(Q), _p(K), _p(V), _p(key_mask),
                                          _p(query_mask), _p(O), _p(P), _st(Q)), "tcar_mha_core_fwd")
    return O, P


@mha_core.register_fake
def _(Q, K, V, key_mask, query_mask, heads, causal):
    return torch.empty_like(Q), Q.new_empty(Q.shape[0] * heads, Q.shape[1], K.shape[1])


@torch.library.custom_op("tcar::mha_core_bwd", mutates_args=(), device_types="cuda")
def mha_core_bwd(Q: Tensor, K: Tensor, V: Tensor, P: Tensor, key_mask: Tensor, query_mask: Tensor, dO: Tensor, heads: int,
                 causal: bool) -> Tuple[Tensor, Tensor, Tensor]:
    _chk(Q, K, V, P, key_mask, query_mask, dO)
    N, Tq, Cc = Q.shape
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
    check(_lib_().tcar_mha_core_bwd_tuned(_tune_ptr(), N, Tq, K.shape[1], Cc, heads, int(causal), _p(Q), _p(K), _p(V), _p(P),
                                          _p(key_mask), _p(query_mask), _p(dO), _p(dQ), _p(dK), _p(dV), _st(Q)), "tcar_mha_core_bwd")
    return dQ, dK, dV


@mha_core_bwd.register_fake
def _(Q, K, V, P, key_mask, query_mask, dO, heads, causal):
    return torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)


def _mha_setup(ctx, inputs, output):
    Q, K, V, km, qm, heads, causal = inputs
    ctx.save_for_backward(Q, K, V, output[1], km, qm)
    ctx.hc = (heads, causal)


def _mha_backward(ctx, dO, _dP):
    Q, K, V, P, km, qm = ctx.saved_tensors
    dQ, dK, dV = torch.ops.tcar.mha_core_bwd(Q, K, V, P, km, qm, dO.contiguous(), ctx.hc[0], ctx.hc[1])
    return dQ, dK, dV, None, None, None, None


mha_core.register_autograd(_mha_backward, setup_context=_mha_setup)


def multihead_attention(queries: Tensor, keys: Tensor, wq: Tensor, bq: Tensor, wk: Tensor, bk: Tensor, wv: Tensor, bv: Tensor,
                        num_heads: int = 8, causality: bool = False) -> Tensor:
    """modules.py:220-304 with dropout off: dense Q / K / V projections without activation (:247-249), head split, the
    attention core (`mha_core`), head merge, residual (:298).  queries [N,Tq,C], keys [N,Tk,C], w* [C,C], b* [C]."""
    N, Tq, Cc = queries.shape
    Tk = keys.shape[1]
    q2, k2 = queries.reshape(N * Tq, Cc).contiguous(), keys.reshape(N * Tk, Cc).contiguous()
    Q = torch.ops.tcar.linear(q2, wq, bq, 0).reshape(N, Tq, Cc)
    K = torch.ops.tcar.linear(k2, wk, bk, 0).reshape(N, Tk, Cc)
    V = torch.ops.tcar.linear(k2, wv, bv, 0).reshape(N, Tk, Cc)
    key_mask = torch.sign(keys.detach().sum(-1).abs()).contiguous()          # modules.py:263
    query_mask = torch.sign(queries.detach().sum(-1).abs()).contiguous()     # modules.py:283
    O, _ = torch.ops.tcar.mha_core(Q, K, V, key_mask, query_mask, num_heads, bool(causality))
    return O + queries


OPS = ("gather_clip", "attn_pool", "score_ce", "score_rank", "neg_term", "clip_adam_", "rank_topk", "linear", "linear_residual",
       "normalize", "mha_core")
