"""The autograd wrappers of tcar_amd.ops against plain PyTorch fp32 restatements of the same formulas (forward values and
every input gradient through torch.autograd)."""
import numpy as np
import pytest
import torch

import tcar_amd  # noqa: F401

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _expnorm(x, dim=1):
    e = torch.exp(x)
    return e / (e.sum(dim, keepdim=True) + 1e-9)            # util.py:92-100


def _pool_ref(x_icp, x_pt, pre1, pre2, q, w1, w2):
    a1 = _expnorm((torch.sigmoid(pre1) * w1).sum(-1))        # modules.py:132-135
    a2 = _expnorm((x_icp * q[:, None, :]).sum(-1))           # modules.py:140-141
    a3 = _expnorm((torch.sigmoid(pre2) * w2).sum(-1))        # modules.py:97-100
    return torch.cat([((a1 + a2)[:, :, None] * x_icp).sum(1), (a3[:, :, None] * x_pt).sum(1)], -1)


@pytest.mark.parametrize("B,T", [(5, 1), (33, 4), (3, 40)])
def test_attn_pool_autograd(B, T):
    _need_gpu()
    from tcar_amd import ops
    g = torch.Generator(device="cuda").manual_seed(B * 100 + T)
    H, ldh, ldt = 250, 256, 64
    r = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.5)
    w1, w2 = r(ldh), r(ldh)
    w1[H:], w2[H:] = 0, 0
    ins = [r(B, T, 2 * ldh), r(B, T, 5 * ldt), r(B, T, ldh), r(B, T, ldh), r(B, 2 * ldh), w1, w2]
    pre_mask = torch.zeros(ldh, device="cuda")
    pre_mask[:H] = 1
    a = [t.clone().requires_grad_(True) for t in ins]
    # the PyTorch restatement runs in fp64: through exp(x) / (sum exp(x) + 1e-9) an fp32 autograd of a length-1 session
    # (alpha = 1 - 1e-5) is cancellation noise, the kernel's closed form de = alpha (dalpha - sum dalpha alpha) is not
    b = [t.double().clone().requires_grad_(True) for t in ins]
    out = ops.attn_pool(*a, H)
    ref = _pool_ref(*b)
    assert torch.allclose(out.double(), ref, rtol=2e-4, atol=2e-5)
    go = torch.randn(out.shape, device="cuda", generator=g)
    out.backward(go)
    ref.backward(go.double())
    for i, (x, y) in enumerate(zip(a, b)):
        gx, gy = x.grad.double(), y.grad
        if i in (2, 3, 5, 6):                                 # padding columns j >= H of pre / w_res are structural zeros
            gx, gy = gx * pre_mask, gy * pre_mask
        if T == 1 and i in (2, 3, 4, 5, 6):
            # everything that flows through the normaliser of a length-1 session (alpha = e / (e + 1e-9)) is ~1e-5 of the
            # direct gradients and pure cancellation noise in fp32 (the kernel's alpha is exactly 1): only its size is checked
            assert float(gx.abs().max()) <= 1e-3 * float(b[0].grad.abs().max()), i
            continue
        atol = 2e-3 * float(gy.abs().max()) + 1e-8
        assert torch.allclose(gx, gy, rtol=2e-2, atol=atol), (i, float((gx - gy).abs().max()))


def test_softmax_ce_autograd():
    _need_gpu()
    from tcar_amd import ops
    B, N, ld = 7, 1003, 1024
    x = (torch.randn(B, ld, device="cuda") * 3)
    lab = torch.randint(0, N, (B,), device="cuda")
    a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ce = ops.softmax_ce(a, lab, N)
    ref = torch.nn.functional.cross_entropy(b[:, :N], lab, reduction="none")
    assert torch.allclose(ce, ref, rtol=1e-5, atol=1e-5)
    w = torch.rand(B, device="cuda")
    (ce * w).sum().backward()
    (ref * w).sum().backward()
    assert torch.allclose(a.grad[:, :N], b.grad[:, :N], rtol=1e-4, atol=1e-6) and float(a.grad[:, N:].abs().max()) == 0.0


@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_autograd(act):
    _need_gpu()
    from tcar_amd import ops
    M, K, N = 77, 64, 132
    x, w, bias = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda") * 0.2, torch.randn(N, device="cuda")
    f = [lambda v: v, torch.relu, torch.tanh][act]
    a = [t.clone().requires_grad_(True) for t in (x, w, bias)]
    b = [t.clone().requires_grad_(True) for t in (x, w, bias)]
    y, ref = ops.linear(a[0], a[1], a[2], act), f(b[0] @ b[1] + b[2])
    assert torch.allclose(y, ref, rtol=1e-4, atol=1e-5)
    go = torch.randn_like(y)
    y.backward(go)
    ref.backward(go)
    for p, q in zip(a, b):
        assert torch.allclose(p.grad, q.grad, rtol=1e-3, atol=1e-4)


def test_rank_topk_op():
    _need_gpu()
    from tcar_amd import ops
    B, N = 6, 5000
    x = torch.randn(B, N + 120, device="cuda")
    lab = torch.randint(0, N, (B,), device="cuda")
    rank, topk = ops.rank_topk(x, lab, N, 20)
    xs = x[:, :N]
    want_rank = (xs > xs.gather(1, lab[:, None])).sum(1) + 1
    assert (rank.long() == want_rank).all()
    assert (topk.long() == xs.topk(20, dim=1).indices).all()          # no ties in random data


def _mha_ref(queries, keys, wq, bq, wk, bk, wv, bv, h, causal):
    """PyTorch fp64 restatement of modules.py:220-304 (dropout off)."""
    N, Tq, C = queries.shape
    Tk = keys.shape[1]
    Q, K, V = queries @ wq + bq, keys @ wk + bk, keys @ wv + bv
    split = lambda x: torch.cat(torch.split(x, C // h, dim=2), dim=0)                 # (h*N, T, C/h)
    Q_, K_, V_ = split(Q), split(K), split(V)
    out = Q_ @ K_.transpose(1, 2) / (C // h) ** 0.5
    km = torch.sign(keys.sum(-1).abs()).repeat(h, 1)[:, None, :].expand(-1, Tq, -1)
    pad = torch.full_like(out, -2.0 ** 32 + 1)
    out = torch.where(km == 0, pad, out)
    if causal:
        tril = torch.tril(torch.ones(Tq, Tk, dtype=out.dtype, device=out.device))[None].expand_as(out)
        out = torch.where(tril == 0, pad, out)
    out = torch.softmax(out, -1)
    qm = torch.sign(queries.sum(-1).abs()).repeat(h, 1)[:, :, None]
    out = (out * qm) @ V_
    out = torch.cat(torch.split(out, N, dim=0), dim=2)
    return out + queries


@pytest.mark.parametrize("N,T,C,h,causal", [(3, 5, 32, 4, False), (2, 40, 256, 8, True), (4, 7, 48, 1, True)])
def test_multihead_attention_optional_op(N, T, C, h, causal):
    _need_gpu()
    from tcar_amd import ops
    g = torch.Generator(device="cuda").manual_seed(N * 1000 + T)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    q, k = r(N, T, C) * 0.5, r(N, T, C) * 0.5
    k[0, T - 1] = 0                                            # a padded key position (key mask = 0)
    q[N - 1, 0] = 0                                            # a padded query position (query mask = 0)
    ws = [r(C, C) * 0.1, r(C) * 0.1, r(C, C) * 0.1, r(C) * 0.1, r(C, C) * 0.1, r(C) * 0.1]
    a = [t.clone().requires_grad_(True) for t in [q, k] + ws]
    b = [t.double().clone().requires_grad_(True) for t in [q, k] + ws]
    out = ops.multihead_attention(*a, num_heads=h, causality=causal)
    ref = _mha_ref(*b, h, causal)
    assert torch.allclose(out.double(), ref, rtol=1e-4, atol=1e-5)
    go = torch.randn(out.shape, device="cuda", generator=g)
    out.backward(go)
    ref.backward(go.double())
    gmax = max(float(y.grad.abs().max()) for y in b)          # the K bias has an exactly-zero gradient (softmax shift invariance)
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.allclose(x.grad.double(), y.grad, rtol=2e-3, atol=2e-5 * gmax), (i, float((x.grad.double() - y.grad).abs().max()))
