"""`torch.ops.tcar.*` (tcar_amd.torch_ops): every registered op against the oracle's pieces (oracle.tcar_oracle.clip_rows /
expnorm and fp64 PyTorch restatements of the reference formulas), forward values AND gradients through torch.autograd;
schema / fake-tensor registration through torch.library.opcheck."""
import math

import numpy as np
import pytest
import torch

import tcar_amd  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def close(got, want, rtol=1e-3, atol_scale=2e-5, name=""):
    got = np.asarray(got.detach().cpu() if torch.is_tensor(got) else got, dtype=np.float64)
    want = np.asarray(want.detach().cpu() if torch.is_tensor(want) else want, dtype=np.float64)
    atol = atol_scale * max(1e-30, float(np.abs(want).max()))
    bad = np.abs(got - want) > atol + rtol * np.abs(want)
    assert not bad.any(), "%s: %d / %d off, max abs err %.3e (max |want| %.3e)" % (
        name, int(bad.sum()), bad.size, float(np.abs(got - want).max()), float(np.abs(want).max()))


def _tables(N, H, Ht, rng, scale=0.09):
    ldh, ldt = (H + 63) // 64 * 64, 64
    ek = 2 * ldh + 5 * ldt
    E = np.zeros((N, ek), np.float32)
    E[:, :H] = rng.standard_normal((N, H)) * scale
    E[:, ldh:ldh + H] = rng.standard_normal((N, H)) * scale
    E[:, 2 * ldh:] = rng.standard_normal((N, 5 * ldt)) * 0.05
    pos = np.zeros((40, ldh), np.float32)
    pos[:, :H] = rng.standard_normal((40, H)) * 0.08
    small = np.zeros((150, ldt), np.float32)
    small[:, :Ht] = rng.standard_normal((150, Ht)) * 0.3
    return E, pos, small, ldh, ldt, ek


def _feed(B, T, N, rng):
    seq = rng.randint(1, N + 1, (B, T))
    pub = [rng.randint(1, v, (B, T)) for v in (13, 32, 8, 25, 61)]
    gap = rng.randint(0, 12, (B, T))                      # incl. the out-of-range bucket 11 (zero row, DESIGN S7)
    cw, ch = rng.randint(0, 7, B), rng.randint(0, 24, B)
    feed = np.concatenate([seq.ravel()] + [p.ravel() for p in pub] + [gap.ravel(), cw, ch]).astype(np.int32)
    return seq, pub, gap, cw, ch, feed


def test_ops_are_registered_and_pass_opcheck():
    _need_gpu()
    from tcar_amd import torch_ops
    for name in torch_ops.OPS:
        assert hasattr(torch.ops.tcar, name), name
    rng = np.random.RandomState(0)
    x = torch.tensor(rng.standard_normal((16, 32)).astype(np.float32), device=DEV, requires_grad=True)
    w = torch.tensor(rng.standard_normal((32, 8)).astype(np.float32), device=DEV, requires_grad=True)
    b = torch.zeros(8, device=DEV, requires_grad=True)
    torch.library.opcheck(torch.ops.tcar.linear.default, (x, w, b, 2), test_utils=("test_schema", "test_faketensor",
                                                                                    "test_autograd_registration"))
    lg = torch.randn(4, 64, device=DEV)
    lab = torch.tensor([1, 5, 7, 63], dtype=torch.int32, device=DEV)
    torch.library.opcheck(torch.ops.tcar.rank_topk.default, (lg, lab, 64, 5), test_utils=("test_schema", "test_faketensor"))


@pytest.mark.parametrize("B,T,H", [(7, 3, 250), (33, 1, 250), (4, 40, 100)])
def test_gather_clip_forward_and_table_gradients(B, T, H):
    _need_gpu()
    from oracle.tcar_oracle import clip_rows
    from tcar_amd import torch_ops  # noqa: F401
    rng = np.random.RandomState(B + T)
    N, Ht = 300, 64
    E, pos, small, ldh, ldt, ek = _tables(N, H, Ht, rng)
    seq, pub, gap, cw, ch, feed = _feed(B, T, N, rng)
    Et, post, smt = (torch.tensor(a, device=DEV, requires_grad=True) for a in (E, pos, small))
    out = torch.ops.tcar.gather_clip(Et, post, smt, torch.tensor(feed, device=DEV), B, T, H, Ht)
    # fp64 restatement: gather, clip every gathered row to norm <= 1 (modules.py:36), concat (model_combine.py:65,84,94,111)
    E64, p64, s64 = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (E, pos, small))
    sq = torch.as_tensor(seq - 1)
    item, cont = clip_rows(E64[sq][..., :ldh]), clip_rows(E64[sq][..., ldh:2 * ldh])
    pidx = torch.arange(T)[None, :].expand(B, T)
    x_icp = torch.cat([item + clip_rows(p64[pidx]), cont], -1).reshape(B * T, 2 * ldh)
    off = (0, 13, 45, 53, 78)
    x_pt = torch.cat([clip_rows(s64[torch.as_tensor(pub[k]) + off[k]]) for k in range(5)], -1).reshape(B * T, 5 * ldt)
    g = torch.as_tensor(gap)
    dur = clip_rows(s64[139 + g.clamp(max=10)]) * (g < 11)[..., None]
    x_act = dur.reshape(B * T, ldt)
    click = torch.cat([clip_rows(s64[45 + torch.as_tensor(cw)]), clip_rows(s64[53 + torch.as_tensor(ch)])], -1)
    want = [x_icp, x_pt, x_act, click]
    for name, a, b in zip(("x_icp", "x_pt", "x_act", "click_t"), out, want):
        close(a, b, name=name)
    ws = [torch.tensor(rng.standard_normal(tuple(o.shape)), dtype=torch.float64) for o in want]
    sum((o.double() * w.to(DEV)).sum() for o, w in zip(out, ws)).backward()
    sum((o * w).sum() for o, w in zip(want, ws)).backward()
    close(Et.grad[:, :ldh], E64.grad[:, :ldh], name="d item table", atol_scale=5e-5)
    assert float(Et.grad[:, ldh:].abs().max()) == 0.0            # content is frozen; time columns are not looked up here
    close(post.grad, p64.grad, name="d position table", atol_scale=5e-5)
    close(smt.grad, s64.grad, name="d time / dwell tables", atol_scale=5e-5)


@pytest.mark.parametrize("B,T", [(5, 1), (17, 6)])
def test_attn_pool_op(B, T):
    _need_gpu()
    from oracle.tcar_oracle import expnorm
    from tcar_amd import torch_ops  # noqa: F401
    rng = np.random.RandomState(B)
    H, ldh, pt = 250, 256, 320
    mk = lambda *s, sc=0.3: rng.standard_normal(s).astype(np.float32) * sc
    arrs = [mk(B, T, 2 * ldh), mk(B, T, pt), mk(B, T, ldh, sc=1.0), mk(B, T, ldh, sc=1.0), mk(B, 2 * ldh, sc=0.1), mk(ldh), mk(ldh)]
    arrs[5][H:] = 0
    arrs[6][H:] = 0
    dev = [torch.tensor(a, device=DEV, requires_grad=True) for a in arrs]
    pooled, alpha = torch.ops.tcar.attn_pool(*dev, H)
    d64 = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in arrs]
    x_icp, x_pt, pre1, pre2, q, w1, w2 = d64
    colmask = (torch.arange(ldh) < H).double()
    a1 = expnorm((torch.sigmoid(pre1) * colmask * w1).sum(-1))        # modules.py:132-135
    a2 = expnorm((x_icp * q[:, None, :]).sum(-1))                     # modules.py:140-141
    a3 = expnorm((torch.sigmoid(pre2) * colmask * w2).sum(-1))        # modules.py:97-100
    want = torch.cat([((a1 + a2)[:, :, None] * x_icp).sum(1), (a3[:, :, None] * x_pt).sum(1)], -1)
    close(pooled, want, name="pooled")
    wgt = torch.tensor(rng.standard_normal(tuple(want.shape)), dtype=torch.float64)
    (pooled.double() * wgt.to(DEV)).sum().backward()
    (want * wgt).sum().backward()
    gmax = max(float(b.grad.abs().max()) for b in d64)
    for name, a, b in zip(("x_icp", "x_pt", "pre1", "pre2", "q", "w1", "w2"), dev, d64):
        # T = 1: alpha = e / (e + 1e-9) ~ 1 whatever the scores, so the score-side gradients are zero up to fp32 rounding
        floor = 3e-6 * gmax / max(1e-30, float(b.grad.abs().max()))
        close(a.grad, b.grad, name="d " + name, atol_scale=max(1e-4, floor))


@pytest.mark.parametrize("B,T,skew", [(64, 5, False), (512, 3, True), (33, 40, False), (512, 7, True), (500, 40, True), (300, 9, False)])
def test_order_fixed_small_table_backward(B, T, skew):
    """tcar_small_tables_bwd_det (one workgroup per destination row, sources in order, clip Jacobian once per row) against
    the atomic form inside tcar_gather_clip_bwd: position / time / dwell gradients and their norm pieces agree to rounding,
    the item rows of the skip_small gather are unchanged, and repeated runs agree bit for bit.  skew: every session shares one
    click time and one publish hour (MIND-like: one table row collects every source).  B * T >= 2,048: the chunked form — every table
    row's sources in chunks of ~1,024 with a workgroup each, one wave per row adding the chunk partials in order (3, 16 and 2 chunks
    here)."""
    _need_gpu()
    import ctypes as C
    from tcar_amd import _lib, torch_ops
    from tcar_amd._lib import Grads
    lib = _lib.load()
    rng = np.random.RandomState(B * 7 + T)
    N, H, Ht = 300, 250, 64
    E, pos, small, ldh, ldt, ek = _tables(N, H, Ht, rng, scale=0.2)
    small *= 3.0                                                   # norms above 1: the clip Jacobian is active
    seq, pub, gap, cw, ch, feed = _feed(B, T, N, rng)
    if skew:
        n = B * T
        feed[4 * n:5 * n] = 7                                      # publish hour
        feed[7 * n:7 * n + B] = 3                                  # click week
        feed[7 * n + B:] = 11                                      # click hour
    Et, post, smt, fd = (torch.tensor(a, device=DEV) for a in (E, pos, small, feed))
    mk = lambda *s: torch.tensor(rng.standard_normal(s).astype(np.float32), device=DEV)
    dx_icp, dx_pt, dx_act, dclick = mk(B * T, 2 * ldh), mk(B * T, 5 * ldt), mk(B * T, ldt), mk(B, 2 * ldt)
    ref = torch.ops.tcar.gather_clip_bwd(Et, post, smt, fd, dx_icp, dx_pt, dx_act, dclick, B, T, H, Ht)     # atomic form
    dims, _, _, _ = torch_ops._geom(Et, post, smt, H, Ht)
    tab, bt = torch_ops._tables(Et, post, smt, ldt), torch_ops._batch(fd, B, T)
    p = lambda t: C.c_void_p(t.data_ptr())
    runs = []
    for _ in range(3):
        g_item, g_pos = torch.zeros(N, ldh, device=DEV), torch.zeros(40, ldh, device=DEV)
        g_small, sqn = torch.zeros(150, ldt, device=DEV), torch.zeros(_lib.NSLOT, device=DEV)
        gr = Grads()
        gr.g_item, gr.g_pos, gr.sqn = g_item.data_ptr(), g_pos.data_ptr(), sqn.data_ptr()
        for k in range(5):
            gr.g_time[k] = g_small.data_ptr() + 4 * torch_ops._ROWOFF[k] * ldt
            gr.slot_time[k] = 2 + k
        gr.g_dur = g_small.data_ptr() + 4 * torch_ops._ROWOFF[5] * ldt
        gr.slot_item, gr.slot_pos, gr.slot_dur = 0, 1, 7
        ws = torch.full((lib.tcar_small_det_ws_floats(),), 9.0, device=DEV)
        assert lib.tcar_small_tables_bwd_det(C.byref(dims), C.byref(tab), C.byref(bt), p(dx_icp), p(dx_pt), p(dx_act), p(dclick),
                                             C.byref(gr), p(ws), None) == 0
        gr.skip_small = 1
        assert lib.tcar_gather_clip_bwd(C.byref(dims), C.byref(tab), C.byref(bt), p(dx_icp), p(dx_pt), p(dx_act), p(dclick),
                                        C.byref(gr), None) == 0
        torch.cuda.synchronize()
        runs.append([t.cpu().numpy() for t in (g_item, g_pos, g_small, sqn)])
    for name, a, b in zip(("item rows", "position", "time / dwell", "norm pieces"), runs[0], ref):
        close(a, b.cpu(), name=name, rtol=2e-4, atol_scale=2e-5)
    for r in runs[1:]:
        for name, x, y in zip(("pos", "small", "sqn"), (r[1], r[2], r[3][1:]), (runs[0][1], runs[0][2], runs[0][3][1:])):
            assert np.array_equal(x, y), name                      # bit for bit (the item rows / slot 0 here still use atomics)


def test_order_fixed_pool_backward_and_column_sums():
    """tcar_attn_pool_bwd_det + tcar_colsum_det (the step driver's form in the split-bf16 modes) against the atomic form
    tcar_attn_pool_bwd_q: same dx / dpre / dq, the column sums of gw_rows and dq equal the residual-weight and bias gradients,
    and three runs agree bit for bit."""
    _need_gpu()
    import ctypes as C
    from tcar_amd import _lib
    from tcar_amd._lib import Dims
    lib = _lib.load()

    class Colsum(C.Structure):
        _fields_ = [("x", C.c_void_p), ("ld", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32), ("dst", C.c_void_p)]
    rng = np.random.RandomState(11)
    B, T, H, ldh, pt = 301, 5, 250, 256, 320
    ic = 2 * ldh
    mk = lambda *s, sc=0.3: torch.tensor(rng.standard_normal(s).astype(np.float32) * sc, device=DEV)
    x_icp, x_pt, pre1, pre2, q = mk(B * T, ic), mk(B * T, pt), mk(B * T, ldh, sc=1.0), mk(B * T, ldh, sc=1.0), torch.tanh(mk(B, ic))
    w1, w2 = mk(ldh), mk(ldh)
    w1[H:] = 0
    w2[H:] = 0
    dpooled = mk(B, ic + pt)
    d = Dims(1, H, 64, ldh, 64)
    p = lambda t: C.c_void_p(t.data_ptr())
    pooled, alpha = torch.empty(B, ic + pt, device=DEV), torch.empty(3, B * T, device=DEV)
    assert lib.tcar_attn_pool_fwd(C.byref(d), B, T, p(x_icp), p(x_pt), p(pre1), p(pre2), p(q), p(w1), p(w2), p(pooled), p(alpha), None) == 0
    outs = lambda: [torch.empty_like(x_icp), torch.empty_like(x_pt), torch.empty_like(q), torch.empty_like(pre1), torch.empty_like(pre2)]
    ref = outs()
    g1, g2, gq = torch.zeros(ldh, device=DEV), torch.zeros(ldh, device=DEV), torch.zeros(ic, device=DEV)
    assert lib.tcar_attn_pool_bwd_q(C.byref(d), B, T, p(x_icp), p(x_pt), p(pre1), p(pre2), p(q), p(w1), p(w2), p(alpha), p(dpooled),
                                    *[p(t) for t in ref], p(g1), p(g2), p(gq), None) == 0
    runs = []
    for _ in range(3):
        got = outs()
        gw = torch.full((B, ic), 7.0, device=DEV)
        assert lib.tcar_attn_pool_bwd_det(C.byref(d), B, T, p(x_icp), p(x_pt), p(pre1), p(pre2), p(q), p(w1), p(w2), p(alpha), p(dpooled),
                                          *[p(t) for t in got], p(gw), None) == 0
        dst = torch.zeros(ic + 2 * ldh, device=DEV)
        segs = (Colsum * 3)()
        for i, (x, ld, cols, off) in enumerate(((got[2], ic, ic, 0), (gw, ic, ldh, ic), (gw[:, ldh:], ic, ldh, ic + ldh))):
            segs[i].x, segs[i].ld, segs[i].rows, segs[i].cols, segs[i].dst = x.data_ptr(), ld, B, cols, dst.data_ptr() + 4 * off
        assert lib.tcar_colsum_det(3, C.cast(segs, C.c_void_p), None) == 0
        torch.cuda.synchronize()
        runs.append([t.cpu().numpy() for t in got] + [dst.cpu().numpy()])
    for a, b in zip(runs[0][:5], ref):
        assert np.array_equal(a, b.cpu().numpy())                       # element-wise outputs: the same arithmetic
    close(torch.tensor(runs[0][5][:ic]), gq.cpu().double(), name="d q-bias", atol_scale=1e-5)
    close(torch.tensor(runs[0][5][ic:ic + ldh]), g1.cpu().double(), name="d w_res1", atol_scale=1e-5)
    close(torch.tensor(runs[0][5][ic + ldh:]), g2.cpu().double(), name="d w_res2", atol_scale=1e-5)
    want = np.concatenate([runs[0][2].astype(np.float64).sum(0)])
    np.testing.assert_allclose(runs[0][5][:ic], want, rtol=1e-5, atol=1e-6)
    for r in runs[1:]:
        assert all(np.array_equal(x, y) for x, y in zip(r, runs[0]))


def test_score_ce_and_score_rank_ops():
    _need_gpu()
    from tcar_amd import torch_ops  # noqa: F401
    rng = np.random.RandomState(3)
    B, N, ek = 37, 1003, 832
    att = torch.tensor(np.tanh(rng.standard_normal((B, ek))).astype(np.float32), device=DEV, requires_grad=True)
    E = torch.tensor((rng.standard_normal((N, ek)) * 0.1).astype(np.float32), device=DEV, requires_grad=True)
    lab_np = rng.randint(0, N, B)
    lab = torch.tensor(lab_np, dtype=torch.int32, device=DEV)
    ce, _ = torch.ops.tcar.score_ce(att, E, lab)
    a64, e64 = (torch.tensor(t.detach().cpu().numpy(), dtype=torch.float64, requires_grad=True) for t in (att, E))
    logits = a64 @ e64.T                                             # model_combine.py:138
    want = torch.nn.functional.cross_entropy(logits, torch.as_tensor(lab_np), reduction="none")    # :145
    close(ce, want, name="ce")
    wgt = torch.tensor(rng.uniform(0.5, 1.5, B))
    (ce.double() * wgt.to(DEV)).sum().backward()
    (want * wgt).sum().backward()
    close(att.grad, a64.grad, name="d attout", atol_scale=5e-5)
    close(E.grad, e64.grad, name="d E", atol_scale=5e-5)
    rank, topk, ce2 = torch.ops.tcar.score_rank(att.detach(), E.detach(), lab, 20)
    lg = logits.detach()
    want_rank = (lg > lg.gather(1, torch.as_tensor(lab_np)[:, None])).sum(1) + 1      # util.py:13-14 (strict >)
    assert (rank.cpu().long() == want_rank).all()
    tv = lg.gather(1, topk.cpu().long())
    assert (tv[:, :-1] >= tv[:, 1:]).all() and torch.allclose(tv[:, -1], lg.topk(20, 1).values[:, -1], rtol=1e-4)
    close(ce2, want, name="eval ce")


def test_neg_term_op():
    _need_gpu()
    from tcar_amd import torch_ops  # noqa: F401
    rng = np.random.RandomState(5)
    B, K, N, H, Ht, ldh = 29, 20, 500, 250, 64, 256
    ek = 2 * ldh + 320
    E_np = np.zeros((N, ek), np.float32)
    E_np[:, :H] = rng.standard_normal((N, H)) * 0.03
    E_np[:, ldh:ldh + H] = rng.standard_normal((N, H)) * 0.03
    att_np = np.tanh(rng.standard_normal((B, ek))).astype(np.float32)
    neg_np = rng.randint(0, N, (B, K))
    E = torch.tensor(E_np, device=DEV, requires_grad=True)
    att = torch.tensor(att_np, device=DEV, requires_grad=True)
    neg = torch.tensor(neg_np, dtype=torch.int32, device=DEV)
    fb, _, _ = torch.ops.tcar.neg_term(E, neg, att, H, Ht)
    e64 = torch.tensor(E_np, dtype=torch.float64, requires_grad=True)
    a64 = torch.tensor(att_np, dtype=torch.float64, requires_grad=True)
    x = (e64[torch.as_tensor(neg_np)][..., :2 * ldh] * a64[:, None, :2 * ldh]).sum((1, 2))     # model_combine.py:142
    want = -torch.log(1 - torch.sigmoid(x) + 1e-24)                                             # :143
    close(fb, want, name="neg_fb")
    wgt = torch.tensor(rng.uniform(0.5, 1.5, B))
    (fb.double() * wgt.to(DEV)).sum().backward()
    (want * wgt).sum().backward()
    close(att.grad, a64.grad, name="d attout", atol_scale=5e-5)
    close(E.grad[:, :ldh], e64.grad[:, :ldh], name="d item columns", atol_scale=5e-5)
    assert float(E.grad[:, ldh:].abs().max()) == 0.0       # content columns are frozen in the reference (model_combine.py:67-68)


def test_clip_adam_op_matches_the_tf1_update():
    _need_gpu()
    from tcar_amd import torch_ops  # noqa: F401
    rng = np.random.RandomState(9)
    n = 4096
    w0, g0 = rng.standard_normal(n).astype(np.float32), (rng.standard_normal(n) * 3).astype(np.float32)
    m0, v0 = (rng.standard_normal(n) * 0.1).astype(np.float32), np.abs(rng.standard_normal(n) * 0.01).astype(np.float32)
    for clip, pieces, dense in ((150.0, 0.0, True), (5.0, 0.0, True), (5.0, 123.0, False), (5.0, 40.0, True)):
        w, g, m, v = (torch.tensor(a, device=DEV) for a in (w0, g0, m0, v0))
        lr_t, b1, b2, eps = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9), 0.9, 0.999, 1e-8
        torch.ops.tcar.clip_adam_(w, g, m, v, pieces, dense, clip, lr_t, b1, b2, eps)
        nrm = math.sqrt((float((g0.astype(np.float64) ** 2).sum()) if dense else 0.0) + pieces)
        gc = g0.astype(np.float64) * (clip / max(nrm, clip))                      # tf.clip_by_norm (model_combine.py:157-160)
        m1 = b1 * m0 + (1 - b1) * gc
        v1 = b2 * v0 + (1 - b2) * gc * gc
        w1 = w0 - lr_t * m1 / (np.sqrt(v1) + eps)                                 # TF-1 Adam (DESIGN S6)
        close(m, m1, name="m")
        close(v, v1, name="v")
        close(w, w1, name="w", atol_scale=1e-6)


def test_rank_topk_and_linear_ops():
    _need_gpu()
    from tcar_amd import torch_ops  # noqa: F401
    rng = np.random.RandomState(11)
    lg = torch.tensor(rng.standard_normal((9, 260)).astype(np.float32), device=DEV)
    lg[3, :] = 0.5                                                   # an all-tie row
    lab = torch.tensor(rng.randint(0, 257, 9), dtype=torch.int32, device=DEV)
    rank, topk = torch.ops.tcar.rank_topk(lg, lab, 257, 20)
    x = lg[:, :257].cpu().numpy()
    for b in range(9):
        assert int(rank[b]) == int((x[b] > x[b, int(lab[b])]).sum()) + 1
        assert topk[b].cpu().numpy().tolist() == np.argsort(x[b], kind="stable")[::-1][:20].tolist()
    xs = torch.tensor(rng.standard_normal((40, 64)).astype(np.float32), device=DEV, requires_grad=True)
    w = torch.tensor((rng.standard_normal((64, 48)) * 0.2).astype(np.float32), device=DEV, requires_grad=True)
    bias = torch.tensor(rng.standard_normal(48).astype(np.float32), device=DEV, requires_grad=True)
    for act, f in ((0, lambda t: t), (1, torch.relu), (2, torch.tanh)):
        y = torch.ops.tcar.linear(xs, w, bias, act)
        x64, w64, b64 = (t.detach().double().cpu().requires_grad_(True) for t in (xs, w, bias))
        want = f(x64 @ w64 + b64)
        close(y, want, name="linear act %d" % act)
        for t in (xs, w, bias):
            t.grad = None
        wgt = torch.tensor(rng.standard_normal((40, 48)))
        (y.double() * wgt.to(DEV)).sum().backward()
        (want * wgt).sum().backward()
        close(xs.grad, x64.grad, name="dx", atol_scale=5e-5)
        close(w.grad, w64.grad, name="dw", atol_scale=5e-5)
        close(bias.grad, b64.grad, name="db", atol_scale=5e-5)


# ------------------------------------------------------------- optional blocks of modules.py:194-336
@pytest.mark.parametrize("shape", [(7, 5, 250), (300, 64), (3, 2, 2048), (5, 33)])
def test_normalize_op(shape):
    _need_gpu()
    from tcar_amd import torch_ops  # noqa: F401
    rng = np.random.RandomState(sum(shape))
    C_ = shape[-1]
    x = torch.tensor((rng.standard_normal(shape) * 2 + 0.5).astype(np.float32), device=DEV, requires_grad=True)
    gamma = torch.tensor((1 + 0.3 * rng.standard_normal(C_)).astype(np.float32), device=DEV, requires_grad=True)
    beta = torch.tensor((0.2 * rng.standard_normal(C_)).astype(np.float32), device=DEV, requires_grad=True)
    y, _ = torch.ops.tcar.normalize(x, gamma, beta, 1e-8)
    x64, g64, b64 = (t.detach().double().cpu().requires_grad_(True) for t in (x, gamma, beta))
    mean, var = x64.mean(-1, keepdim=True), x64.var(-1, unbiased=False, keepdim=True)       # tf.nn.moments (modules.py:213)
    want = g64 * (x64 - mean) / (var + 1e-8) ** 0.5 + b64                                   # modules.py:216-217
    close(y, want, name="normalize")
    wgt = torch.tensor(rng.standard_normal(shape))
    (y.double() * wgt.to(DEV)).sum().backward()
    (want * wgt).sum().backward()
    close(x.grad, x64.grad, name="dx", atol_scale=1e-4)
    close(gamma.grad, g64.grad, name="dgamma", atol_scale=1e-4)
    close(beta.grad, b64.grad, name="dbeta", atol_scale=1e-4)


def test_feedforward_block():
    _need_gpu()
    from tcar_amd import torch_ops
    rng = np.random.RandomState(21)
    N, T, C_, F = 6, 9, 64, 256
    mk = lambda *s, sc=0.2: torch.tensor((rng.standard_normal(s) * sc).astype(np.float32), device=DEV, requires_grad=True)
    x, w1, b1, w2, b2 = mk(N, T, C_, sc=1.0), mk(C_, F), mk(F), mk(F, C_), mk(C_)
    y = torch_ops.feedforward(x, w1, b1, w2, b2)
    d = [t.detach().double().cpu().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    want = torch.relu(d[0] @ d[1] + d[2]) @ d[3] + d[4] + d[0]                               # modules.py:319-333
    close(y, want, name="feedforward")
    wgt = torch.tensor(rng.standard_normal((N, T, C_)))
    (y.double() * wgt.to(DEV)).sum().backward()
    (want * wgt).sum().backward()
    for name, a, b in zip(("x", "w1", "b1", "w2", "b2"), (x, w1, b1, w2, b2), d):
        close(a.grad, b.grad, name="d " + name, atol_scale=1e-4)


def _mha_ref(queries, keys, wq, bq, wk, bk, wv, bv, h, causal):
    N, Tq, C_ = queries.shape
    Tk = keys.shape[1]
    Q, K, V = queries @ wq + bq, keys @ wk + bk, keys @ wv + bv                              # modules.py:247-249
    split = lambda t: torch.cat(torch.split(t, C_ // h, dim=2), dim=0)                       # :252-254
    Q_, K_, V_ = split(Q), split(K), split(V)
    out = Q_ @ K_.transpose(1, 2) / (C_ // h) ** 0.5                                         # :257-261
    km = torch.sign(keys.sum(-1).abs()).repeat(h, 1)[:, None, :].expand(-1, Tq, -1)          # :263-265
    out = torch.where(km == 0, torch.full_like(out, -2.0 ** 32 + 1), out)                    # :267-268
    if causal:
        tril = torch.tril(torch.ones(Tq, Tk, dtype=out.dtype))
        out = torch.where(tril[None] == 0, torch.full_like(out, -2.0 ** 32 + 1), out)        # :271-277
    out = torch.softmax(out, -1)                                                             # :280
    qm = torch.sign(queries.sum(-1).abs()).repeat(h, 1)[:, :, None]                          # :283-285
    out = (out * qm) @ V_                                                                    # :286,292
    return torch.cat(torch.split(out, N, dim=0), dim=2) + queries                            # :295-298


@pytest.mark.parametrize("N,T,C_,h,causal", [(5, 40, 256, 8, True), (3, 17, 128, 2, False), (4, 64, 64, 2, True),
                                               (2, 9, 48, 3, False)])
def test_multihead_attention_block_on_the_matrix_cores(N, T, C_, h, causal):
    """head sizes 32 and 64 run the MFMA form, 16 the scalar one; padded keys / queries (all-zero rows) exercise the key and
    query masks, and one sample has EVERY key masked (the reference's softmax is then uniform over the masked keys)"""
    _need_gpu()
    from tcar_amd import torch_ops
    rng = np.random.RandomState(N * 1000 + T)
    mk = lambda *s, sc=0.3: (rng.standard_normal(s) * sc).astype(np.float32)
    q_np, k_np = mk(N, T, C_, sc=1.0), mk(N, T, C_, sc=1.0)
    k_np[0, T // 2:] = 0          # padded keys
    q_np[1, -2:] = 0              # padded queries
    k_np[N - 1] = 0               # every key masked
    ws = [mk(C_, C_, sc=C_ ** -0.5), mk(C_, sc=0.1), mk(C_, C_, sc=C_ ** -0.5), mk(C_, sc=0.1), mk(C_, C_, sc=C_ ** -0.5), mk(C_, sc=0.1)]
    dev = [torch.tensor(a, device=DEV, requires_grad=True) for a in [q_np, k_np] + ws]
    y = torch_ops.multihead_attention(*dev, num_heads=h, causality=causal)
    d = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in [q_np, k_np] + ws]
    want = _mha_ref(*d, h, causal)
    close(y, want, name="multihead_attention", atol_scale=1e-4)
    wgt = torch.tensor(rng.standard_normal((N, T, C_)))
    (y.double() * wgt.to(DEV)).sum().backward()
    (want * wgt).sum().backward()
    gmax = max(float(b.grad.abs().max()) for b in d)
    for name, a, b in zip(("queries", "keys", "wq", "bq", "wk", "bk", "wv", "bv"), dev, d):
        # d bk is zero in exact arithmetic (a bias on the keys shifts every score of a query alike: softmax is invariant)
        floor = 3e-6 * gmax / max(1e-30, float(b.grad.abs().max()))
        close(a.grad, b.grad, name="d " + name, atol_scale=max(2e-4, floor))
    if C_ // h in (32, 64):
        # the MFMA form against the scalar form of the same entry point
        from tcar_amd import _lib as _lib_mod
        Q = torch.tensor(mk(N, T, C_), device=DEV)
        Kt, V = torch.tensor(mk(N, T, C_), device=DEV), torch.tensor(mk(N, T, C_), device=DEV)
        km = torch.sign(torch.tensor(k_np, device=DEV).sum(-1).abs()).contiguous()
        qm = torch.sign(torch.tensor(q_np, device=DEV).sum(-1).abs()).contiguous()
        outs = []
        for flag in (1, 0):
            torch_ops.TUNING = _lib_mod.tuning(TCAR_MHA_MFMA=flag)
            try:
                O, P = torch.ops.tcar.mha_core(Q, Kt, V, km, qm, h, causal)
                g = torch.ops.tcar.mha_core_bwd(Q, Kt, V, P, km, qm, torch.ones_like(O), h, causal)
                torch.cuda.synchronize()
            finally:
                torch_ops.TUNING = None
            outs.append([O, P] + list(g))
        for name, a, b in zip(("O", "P", "dQ", "dK", "dV"), outs[0], outs[1]):
            close(a, b, name="mfma vs scalar " + name, atol_scale=1e-5)
