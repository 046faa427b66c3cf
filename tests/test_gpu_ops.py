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
