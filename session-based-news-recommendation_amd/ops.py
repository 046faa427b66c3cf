"""PyTorch autograd wrappers over the op-level C-ABI ("op-level boundary"): the entry points of
include/tcar_hip.h whose forward / backward are self-contained take and return plain tensors here, so they compose with
ordinary autograd code.  They are thin: every arithmetic operation happens in libtcar_hip.so on the CURRENT torch stream;
tensors must be fp32, contiguous and on the GPU (no CPU fallback: `_lib.load` raises without the library).

    attn_pool(x_icp, x_pt, pre1, pre2, q, w_res1, w_res2, H) -> pooled          modules.py:72-152, util.py:92-100
    softmax_ce(logits, label, n_valid)                      -> ce [B]            model_combine.py:145
    linear(x, w, bias=None, act=0)                          -> act(x @ w + b)    modules.py:43-70 (fp32 MFMA)
    rank_topk(logits, label, n_valid, k)                    -> rank, topk        util.py:13-17, model_combine.py:301

The training path proper does not go through these (it is sequenced by the C++ step driver, csrc/step.hip); they are the
binding a maintainer would use to call single ops from a PyTorch model, and what tests/test_gpu_ops.py checks against
plain PyTorch fp32 restatements of the same formulas.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import Dims, check


def _p(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


def _st(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.TcarError("tcar ops need CUDA (HIP) tensors: there is no CPU fallback")
    return t.contiguous().float()


class _AttnPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_icp, x_pt, pre1, pre2, q, w1, w2, H: int):
        lib = _lib.load()
        x_icp, x_pt, pre1, pre2, q, w1, w2 = map(_f32c, (x_icp, x_pt, pre1, pre2, q, w1, w2))
        B, T, ic = x_icp.shape
        ldh, pt = ic // 2, x_pt.shape[2]
        dims = Dims(1, H, pt // 5, ldh, pt // 5)
        pooled = torch.empty(B, ic + pt, device=x_icp.device)
        alpha = torch.empty(3, B * T, device=x_icp.device)
        check(lib.tcar_attn_pool_fwd(C.byref(dims), B, T, _p(x_icp), _p(x_pt), _p(pre1), _p(pre2), _p(q), _p(w1), _p(w2),
                                     _p(pooled), _p(alpha), _st(x_icp)), "tcar_attn_pool_fwd")
        ctx.save_for_backward(x_icp, x_pt, pre1, pre2, q, w1, w2, alpha)
        ctx.dims = dims
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        lib = _lib.load()
        x_icp, x_pt, pre1, pre2, q, w1, w2, alpha = ctx.saved_tensors
        B, T, ic = x_icp.shape
        dpooled = _f32c(dpooled)
        dx_icp, dx_pt = torch.empty_like(x_icp), torch.empty_like(x_pt)
        dq, dpre1, dpre2 = torch.empty_like(q), torch.empty_like(pre1), torch.empty_like(pre2)
        g1, g2 = torch.zeros_like(w1), torch.zeros_like(w2)
        check(lib.tcar_attn_pool_bwd(C.byref(ctx.dims), B, T, _p(x_icp), _p(x_pt), _p(pre1), _p(pre2), _p(q), _p(w1), _p(w2),
                                     _p(alpha), _p(dpooled), _p(dx_icp), _p(dx_pt), _p(dq), _p(dpre1), _p(dpre2), _p(g1),
                                     _p(g2), _st(x_icp)), "tcar_attn_pool_bwd")
        return dx_icp, dx_pt, dpre1, dpre2, dq, g1, g2, None


def attn_pool(x_icp, x_pt, pre1, pre2, q, w_res1, w_res2, H: int):
    """x_icp [B,T,2*ldh] (item | content), x_pt [B,T,5*ldt], pre1 / pre2 [B,T,ldh] (pre-sigmoid attention features), q
    [B,2*ldh], w_res1 / w_res2 [ldh] (zero beyond H) -> pooled [B, 2*ldh + 5*ldt]."""
    return _AttnPool.apply(x_icp, x_pt, pre1, pre2, q, w_res1, w_res2, H)


class _SoftmaxCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, label, n_valid: int):
        lib = _lib.load()
        work = _f32c(logits).clone()                 # the kernel overwrites its input with softmax - onehot
        B, ld = work.shape
        lab = label.to(torch.int32).contiguous()
        ce = torch.empty(B, device=work.device)
        check(lib.tcar_softmax_ce(B, n_valid, _p(work), ld, _p(lab), _p(ce), _st(work)), "tcar_softmax_ce")
        ctx.save_for_backward(work)
        return ce

    @staticmethod
    def backward(ctx, dce):
        (dlogits,) = ctx.saved_tensors
        return dlogits * dce[:, None], None, None


def softmax_ce(logits, label, n_valid=None):
    """sparse softmax cross entropy over the first n_valid columns of logits [B, ld] (ld % 4 == 0)."""
    return _SoftmaxCE.apply(logits, label, logits.shape[1] if n_valid is None else n_valid)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, act: int):
        lib = _lib.load()
        x, w = _f32c(x), _f32c(w)
        M, K = x.shape
        N = w.shape[1]
        y = torch.empty(M, N, device=x.device)
        b = _f32c(bias) if bias is not None else None
        check(lib.tcar_gemm_f32(0, M, N, K, _p(x), K, _p(w), N, _p(y), N, _p(b) if b is not None else None, act, 0, 1,
                                _st(x)), "tcar_gemm_f32")
        ctx.save_for_backward(x, w, y)
        ctx.act, ctx.has_bias = act, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, y = ctx.saved_tensors
        dz = _f32c(dy).clone()
        M, K = x.shape
        N = w.shape[1]
        db = torch.zeros(N, device=x.device)
        # dz = dy * act'(y) in place, db = column sums (tcar_dact_colsum), then dx = dz w^T and dw = x^T dz
        check(lib.tcar_dact_colsum(M, N, N, _p(y), _p(dz), _p(db), ctx.act, _st(x)), "tcar_dact_colsum")
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        check(lib.tcar_gemm_f32(1, M, K, N, _p(dz), N, _p(w), N, _p(dx), K, None, 0, 0, 1, _st(x)), "tcar_gemm_f32")
        check(lib.tcar_gemm_f32(2, K, N, M, _p(x), K, _p(dz), N, _p(dw), N, None, 0, 0, 1, _st(x)), "tcar_gemm_f32")
        return dx, dw, (db if ctx.has_bias else None), None


def linear(x, w, bias=None, act: int = 0):
    """act(x @ w + bias) on the fp32 MFMA GEMM; x [M,K], w [K,N] (K, N % 4 == 0), act: 0 none, 1 relu, 2 tanh."""
    return _Linear.apply(x, w, bias, act)


def rank_topk(logits, label, n_valid=None, k: int = 20):
    """rank[b] = 1 + #{n: logits[b,n] > logits[b,label[b]]} and the top-k indices in np.argsort(x)[::-1] order."""
    lib = _lib.load()
    x = _f32c(logits)
    B, ld = x.shape
    n = ld if n_valid is None else n_valid
    lab = label.to(torch.int32).contiguous()
    rank = torch.empty(B, dtype=torch.int32, device=x.device)
    topk = torch.empty(B, k, dtype=torch.int32, device=x.device)
    check(lib.tcar_rank_topk(B, n, _p(x), ld, _p(lab), k, _p(rank), _p(topk), _st(x)), "tcar_rank_topk")
    return rank, topk


# ---- optional: multihead_attention (modules.py:220-304) — not on TCAR's executed graph ---------------------------------
class _MhaCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Q, K, V, key_mask, query_mask, heads: int, causal: bool):
        lib = _lib.load()
        Q, K, V, km, qm = map(_f32c, (Q, K, V, key_mask, query_mask))
        N, Tq, Cc = Q.shape
        Tk = K.shape[1]
        O = torch.empty_like(Q)
        P = torch.empty(N * heads, Tq, Tk, device=Q.device)
        check(lib.tcar_mha_core_fwd(N, Tq, Tk, Cc, heads, int(causal), _p(Q), _p(K), _p(V), _p(km), _p(qm), _p(O), _p(P),
                                    _st(Q)), "tcar_mha_core_fwd")
        ctx.save_for_backward(Q, K, V, P, km, qm)
        ctx.heads, ctx.causal = heads, int(causal)
        return O

    @staticmethod
    def backward(ctx, dO):
        lib = _lib.load()
        Q, K, V, P, km, qm = ctx.saved_tensors
        N, Tq, Cc = Q.shape
        dO = _f32c(dO)
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
        check(lib.tcar_mha_core_bwd(N, Tq, K.shape[1], Cc, ctx.heads, ctx.causal, _p(Q), _p(K), _p(V), _p(P), _p(km), _p(qm),
                                    _p(dO), _p(dQ), _p(dK), _p(dV), _st(Q)), "tcar_mha_core_bwd")
        return dQ, dK, dV, None, None, None, None


def multihead_attention(queries, keys, wq, bq, wk, bk, wv, bv, num_heads=8, causality=False):
    """modules.py:220-304 with dropout off: dense Q/K/V projections without activation (:247-249), head split, scaled
    dot-product scores with the key mask (:263-268) and the optional causal mask (:271-277), softmax, query mask
    (:283-286), weighted sum, head merge, residual (:298).  queries [N,Tq,C], keys [N,Tk,C], w* [C,C], b* [C]."""
    N, Tq, Cc = queries.shape
    Tk = keys.shape[1]
    Q = linear(queries.reshape(N * Tq, Cc), wq, bq).reshape(N, Tq, Cc)
    K = linear(keys.reshape(N * Tk, Cc), wk, bk).reshape(N, Tk, Cc)
    V = linear(keys.reshape(N * Tk, Cc), wv, bv).reshape(N, Tk, Cc)
    key_mask = torch.sign(keys.detach().sum(-1).abs())
    query_mask = torch.sign(queries.detach().sum(-1).abs())
    return _MhaCore.apply(Q, K, V, key_mask, query_mask, num_heads, bool(causality)) + queries
