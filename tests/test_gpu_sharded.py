"""Catalog-sharded data-parallel step (tcar_amd.sharded.ShardedEngine).

* world 1: the Python-sequenced sharded path (per-shard softmax statistics, combine, range scatters, shard ctx update) against
  the fp64 oracle — loss, every gradient, the clip norms, the variables after Adam steps;
* world 2 on ONE GPU (gloo moving CUDA tensors through the host; RCCL needs > 1 GPU): uneven session shards and a rank whose
  shard of a batch is empty, against a single engine that sees the whole batch; the replicas must agree BIT FOR BIT (every item
  row has one owner, the dense-weight norms are summed in a fixed order);
* the packed exchange rows (tcar_shard_pack_head / _unpack_head / _pack_ids, tcar_scatter_add_rows_packed) and the
  fixed-order dense norm at op level."""
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


@pytest.mark.parametrize("K,scoring", [(7, "bf16x3"), (0, "bf16x3"), (7, "bf16x3-mixed"), (0, "bf16x3-mixed")])
def test_single_rank_sharded_path_matches_oracle(K, scoring):
    """bf16x3: materialised logits of the shard; bf16x3-mixed (the benchmarked precision): the shard runs the single-GPU schedule —
    softmax epilogue with a label window, one-hot time segment, (q, z) form of dE, dP form of dX — between the same exchanges"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tcar_amd  # noqa: F401
    from oracle.tcar_oracle import TcarOracle
    from tcar_amd.sharded import ShardedEngine
    from test_gpu_parity import _case, check_grads, close
    N, H, Ht, B, T = 1000, 250, 64, 33, 5
    params, content, mw, batch = _case(N, H, Ht, B, T, max(K, 1), seed=321)
    if K == 0:
        batch = {k: v for k, v in batch.items() if k != "neg"}
    mixed = scoring == "bf16x3-mixed"
    eng = ShardedEngine(params, content, mw, max_grad=2.0, scoring=scoring, world=1, rank=0)
    ora = TcarOracle(params, content, mw, max_grad=2.0)
    loss = eng.loss_and_grads(batch, cap=B + 3)                      # padded session capacity, as an uneven shard has
    assert eng.onehot == mixed and (eng.s_logits is None) == mixed   # the one-hot schedule keeps no fp32 logits
    o, g_o, sq_o = ora.loss_and_grads(batch)
    close(loss.cpu().numpy(), o["loss"].detach().numpy(), name="loss")
    check_grads(eng.export_grads(), eng.export_sqnorms(), {k: v.numpy() for k, v in g_o.items()}, sq_o, scoring)
    for i in range(2):
        close(eng.train_step(batch).cpu().numpy(), ora.train_step(batch).numpy(), name="train loss", rtol=1e-2 if (mixed and i) else 1e-3)
    p_e, p_o = eng.export_params(), ora.export()
    for k in p_o:      # two Adam steps at lr 1e-3: a rounding-level gradient coordinate may move by up to ~2e-3 * |w| scale
        if mixed:      # (bound of test_gpu_parity.test_step_matches_oracle: bf16 noise may flip a coordinate's direction)
            d = np.abs(p_e[k] - p_o[k]).max()
            assert d <= 1e-3 * np.abs(p_o[k]).max() + 2.0 * 1e-3 * 2, ("param " + k, d)
        else:
            close(p_e[k], p_o[k], name="param " + k, atol_scale=1e-3)
    rank, topk, ce, logits = eng.eval_step(batch, keep_logits=True)
    lo, ce_o = ora.eval_batch(batch)
    # (after two Adam steps: in mixed mode the variables carry the bf16 noise of the scoring gradients, bounded above)
    close(logits.cpu().numpy(), lo.numpy(), name="eval logits", atol_scale=2e-3 if mixed else 1e-4, rtol=1e-2 if mixed else 1e-3)
    close(ce.cpu().numpy(), ce_o.numpy(), name="eval ce", rtol=1e-2 if mixed else 1e-3)


@pytest.mark.parametrize("scoring", ["bf16x3-mixed", "bf16x3"])
def test_single_rank_sharded_split_update_matches_the_update_inside_the_step(scoring):
    """ONE rank: train_step(defer_update=True) owes each step's update to the next step's tcar_shard_begin — early pass (arena + the
    rows that batch gathers) on the main stream, the rest of the table and the arena zero on the aux stream beside the session
    forward — or to flush().  Same arithmetic as the update inside the step: losses of every step, variables and Adam moments agree
    (to the float-atomic noise of this path) over steps with different batches (different rows marked), a flush in the middle, and a
    step without negatives."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tcar_amd  # noqa: F401
    from tcar_amd.sharded import ShardedEngine
    from test_gpu_parity import _case
    N, H, Ht, B, K = 3000, 250, 64, 48, 5
    params, content, mw, b0 = _case(N, H, Ht, B, 3, K, seed=11)
    batches = [b0] + [_case(N, H, Ht, B, T, K, seed=12 + i)[3] for i, T in enumerate((1, 3, 6, 2))]
    batches.append({k: v for k, v in batches[1].items() if k != "neg"})
    a = ShardedEngine(params, content, mw, max_grad=2.0, scoring=scoring, world=1, rank=0)
    b = ShardedEngine(params, content, mw, max_grad=2.0, scoring=scoring, world=1, rank=0)
    e = ShardedEngine(params, content, mw, max_grad=2.0, scoring=scoring, world=1, rank=0)
    e.set_tuning(TCAR_FLAG_FORK=0)                     # the same with every fork and join of the pieces through events
    assert b.can_defer
    la, lb, le = [], [], []
    for i, bt in enumerate(batches):
        la.append(a.train_step(bt).clone())
        lb.append(b.train_step(bt, defer_update=True).clone())
        le.append(e.train_step(bt, defer_update=True).clone())
        assert b._pending_lr is not None
        if i == 2:
            b.flush()                                  # (an update applied by flush: the next step has nothing pending)
            assert b._pending_lr is None
    assert a.step == b.step == e.step == len(batches)
    # an entry point that reads the variables applies the owed update first: evaluation with an update pending
    assert b._pending_lr is not None
    ra, rb = a.eval_step(batches[0]), b.eval_step(batches[0])
    assert b._pending_lr is None
    assert float((ra[2] - rb[2]).abs().max()) <= 1e-4 * float(ra[2].abs().max())          # per-session CE
    assert float((ra[0] != rb[0]).float().mean()) <= 0.02                                  # ranks of the labels (ties at rounding level)
    for x, y in zip(la, le):
        assert float((x - y).abs().max()) <= 1e-4 * float(x.abs().max())
    e.flush()
    for name, x, y in (("M", a.M, e.M), ("V", a.V, e.V), ("Mi", a.Mi, e.Mi), ("Vi", a.Vi, e.Vi)):
        assert float((x - y).abs().max()) <= 5e-3 * float(x.abs().max()), ("events", name)
    # (a and b fork through device flags: slab reduce -> dP readers, input gradients -> weight gradients, gather -> click query -> pools)
    assert b._sig is not None and e.tune is not None
    b.check_forks()
    pa, pb = a.export_params(), b.export_params()      # (export flushes)
    assert b._pending_lr is None
    # Same arithmetic per element.  This path keeps two float-atomic sums (small tables, scatter of the gathered rows), so two runs
    # agree to rounding noise only, and Adam turns noise on a near-zero gradient coordinate into up to lr of weight: the weights are
    # held to the Adam bound of the other tests, the discriminating check is on the MOMENTS, which are linear (m) / quadratic (v) in
    # the gradients — a row that missed one of the six updates, or took one twice, is off by >= 10 % there
    for i, (x, y) in enumerate(zip(la, lb)):
        assert float((x - y).abs().max()) <= 1e-4 * float(x.abs().max()), ("loss of step", i)
    for k in pa:
        assert np.abs(pa[k] - pb[k]).max() <= 1e-3 * np.abs(pa[k]).max() + 0.25 * 1e-3 * len(batches), k
    for name, x, y in (("M", a.M, b.M), ("V", a.V, b.V), ("Mi", a.Mi, b.Mi), ("Vi", a.Vi, b.Vi)):
        assert float((x - y).abs().max()) <= 5e-3 * float(x.abs().max()), name
    rows = (a.Vi - b.Vi).abs().amax(1) / a.Vi.abs().amax(1).clamp_min(1e-30)      # per item row: v is a sum of g^2 terms
    assert float(rows.max()) <= 2e-2, int(rows.argmax())
    assert int(b.adam_bitmap.abs().sum()) == 0          # the marks of the last split update were cleared
    # more than one rank: the switch is accepted and ignored (the owned rows are exchanged behind the update)
    b.world = 2
    assert not b.can_defer
    b.world = 1


def test_single_rank_sharded_anchored_form_matches_oracle():
    """Round 6: the shard pieces take the ANCHORED softmax form (here 253 sessions at a capacity of 256: three padding sessions,
    whose gradient rows stay exactly zero).  Every shard computes the same
    anchor from the gathered attout rows, the statistics exchange adds plain sums, no pass rescales the shard's plane: loss, all 23
    gradients + clip norms against the fp64 oracle at the mixed-precision gate, two training steps, and agreement with the
    group-maximum form (TCAR_FUSED_CE = 1) of the same engine class."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tcar_amd  # noqa: F401
    from oracle.tcar_oracle import TcarOracle
    from tcar_amd.sharded import ShardedEngine
    from test_gpu_parity import _case, check_grads, close, rel_norm
    N, H, Ht, B, T, K = 3000, 250, 64, 253, 3, 7
    params, content, mw, batch = _case(N, H, Ht, B, T, K, seed=323)
    ora = TcarOracle(params, content, mw, max_grad=2.0)
    o, g_o, sq_o = ora.loss_and_grads(batch)
    g_o = {k: v.numpy() for k, v in g_o.items()}
    grads = {}
    for f in (2, 1):
        eng = ShardedEngine(params, content, mw, max_grad=2.0, scoring="bf16x3-mixed", world=1, rank=0)
        eng.set_tuning(TCAR_FUSED_CE=f)
        loss = eng.loss_and_grads(batch, cap=256)
        assert eng.shard_form(256) == {"onehot": True, "ce_anchored": f == 2}
        close(loss.cpu().numpy(), o["loss"].detach().numpy(), name="loss")
        grads[f] = eng.export_grads()
        check_grads(grads[f], eng.export_sqnorms(), g_o, sq_o, "bf16x3-mixed")
        if f == 2:
            for i in range(2):
                close(eng.train_step(batch, cap=256).cpu().numpy(), ora.train_step(batch).numpy(), name="train loss", rtol=1e-2 if i else 1e-3)
        del eng
    for k in g_o:      # the two forms differ by bf16 roundings of the softmax gradient only
        assert rel_norm(grads[2][k], grads[1][k]) <= 5e-3 or np.abs(g_o[k]).max() < 1e-9, (k, rel_norm(grads[2][k], grads[1][k]))


def _worker(rank, world, port, ret, scoring="bf16x3", B=37):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tcar_amd  # noqa: F401
        from tcar_amd.dp import shard_bounds
        from tcar_amd.engine import TcarEngine
        from tcar_amd.sharded import ShardedEngine
        from test_gpu_parity import _case
        N, H, Ht, T, K = 1000, 250, 64, 4, 6                  # B = 37 sessions: uneven shards (19 + 18); 256: two whole 128-row blocks
        params, content, mw, batch = _case(N, H, Ht, B, T, K, seed=5)
        _, _, _, tiny = _case(N, H, Ht, 1, 2, K, seed=6)      # a batch of ONE session: rank 1's shard is empty
        eng = ShardedEngine(params, content, mw, max_grad=2.0, group=dist.group.WORLD, scoring=scoring)
        assert eng.S == 512 and eng.nl == (512 if rank == 0 else 488)
        for step in range(4):
            full = tiny if step == 2 else batch
            b, t = full["seq"].shape
            lo, hi, cap = shard_bounds(b, world, rank)
            sub = {k: v[lo:hi] for k, v in full.items()} if hi > lo else None
            eng.train_step(sub, cap=cap, T=t, K=K)
        torch.cuda.synchronize()
        if scoring == "bf16x3-mixed":       # (the anchored softmax form on both shards, rank 1's at catalog row 512; 19 + 18 or 128 + 128 sessions)
            assert eng.shard_form(shard_bounds(B, world, rank)[2]) == {"onehot": True, "ce_anchored": True}
        got = eng.export_params()
        if rank == 0:
            ref = TcarEngine(params, content, mw, max_grad=2.0, scoring=scoring)
            for step in range(4):
                ref.train_step(tiny if step == 2 else batch)
            want = ref.export_params()
            for k in want:
                d = np.abs(got[k] - want[k]).max()
                # same kernels, different summation order (float atomics, reduction trees): Adam bound as in test_gpu_parity.  (Anchored
                # softmax form, B = 256: the shard's anchor is one wave's dot, the single engine's eight partial dots — the two planes
                # of exponentials are rounded to bf16 relative to slightly different references: the mixed precision's noise bound)
                travel = 2.0 if scoring == "bf16x3-mixed" else 0.25
                assert d <= 1e-3 * np.abs(want[k]).max() + travel * 1e-3 * 4, (k, d)
        flat = torch.cat([torch.tensor(v).reshape(-1) for v in got.values()])
        other = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        assert torch.equal(other[0], other[1]), float((other[0] - other[1]).abs().max())     # replicas: bit for bit
        info = eng.exchange_info()
        assert info["mode"] == "sharded" and info["bytes_per_step"]["item_rows"] == 4 * 2 * 512 * 256
        # checkpoint round trip (host/model.py: save() on every rank, rank 0 writes): export_state is a collective that gathers
        # the owners' Adam moments of the item table; load_state keeps each rank's rows; a resumed engine continues the run
        st = eng.export_state()
        assert st["m/item_emb"].shape == (N + 1, H) and st["v/item_emb"].shape == (N + 1, H)
        assert np.abs(st["v/item_emb"][1:513]).max() > 0 and np.abs(st["v/item_emb"][513:]).max() > 0      # both shards are there
        if rank == 0:
            want_st = ref.export_state()
            for k in ("m/item_emb", "v/item_emb", "m/attout_item_cont_trans/w1"):
                assert np.abs(st[k] - want_st[k]).max() <= (2e-2 if travel > 1 else 2e-3) * np.abs(want_st[k]).max() + 1e-12, k
        eng2 = ShardedEngine(params, content, mw, max_grad=2.0, group=dist.group.WORLD, scoring=scoring)
        eng2.load_state(st)
        assert eng2.step == eng.step and torch.equal(eng2.Mi, eng.Mi) and torch.equal(eng2.Vi, eng.Vi) and torch.equal(eng2.M, eng.M)
        lo, hi, cap = shard_bounds(B, world, rank)
        sub = {k: v[lo:hi] for k, v in batch.items()}
        eng.train_step(sub, cap=cap, T=T, K=K)
        eng2.train_step(sub, cap=cap, T=T, K=K)
        a, b2 = eng.export_params(), eng2.export_params()
        for k in a:      # same state, same step: equal up to the float-atomic order inside a step (gathered-row scatter, small tables)
            assert np.abs(a[k] - b2[k]).max() <= 1e-5 * max(np.abs(a[k]).max(), 1e-3), k
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "FAIL: " + repr(e) + "\n" + traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scoring,B", [("bf16x3", 37), ("bf16x3-mixed", 37), ("bf16x3-mixed", 256)])
def test_two_ranks_sharded_match_single_engine(scoring, B):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.get_context("spawn").Manager()      # (a SPAWNED server: a fork of this process would inherit its GPU state)
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, scoring, B), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)


def test_packed_exchange_rows_and_fixed_order_norm():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctypes as C
    import tcar_amd  # noqa: F401
    from tcar_amd import _lib
    from tcar_amd._lib import Dims, Segments
    lib = _lib.load()
    p = lambda t: C.c_void_p(t.data_ptr())
    rng = np.random.RandomState(5)
    B, cap, ek, K, Kc = 37, 48, 832, 7, 9
    ld = (ek + 2 + Kc + 3) // 4 * 4
    att = torch.tensor(rng.standard_normal((B, ek)).astype(np.float32), device="cuda")
    lab = torch.tensor(rng.randint(0, 1000, B).astype(np.int32), device="cuda")
    coef = torch.tensor(rng.standard_normal(B).astype(np.float32), device="cuda")
    neg = torch.tensor(rng.randint(0, 1000, (B, K)).astype(np.int32), device="cuda")
    head = torch.full((cap, ld), 7.0, device="cuda")
    assert lib.tcar_shard_pack_head(B, cap, ek, K, Kc, p(att), p(lab), p(coef), p(neg), p(head), ld, None) == 0
    hi = head.view(torch.int32)
    assert torch.equal(head[:B, :ek], att) and (head[B:, :ek] == 0).all()
    assert torch.equal(hi[:B, ek], lab) and (hi[B:, ek] == -1).all()
    assert torch.equal(head[:B, ek + 1], coef) and (head[B:, ek + 1] == 0).all()
    assert torch.equal(hi[:B, ek + 2:ek + 2 + K], neg) and (hi[:, ek + 2 + K:ek + 2 + Kc] == -1).all() and (hi[B:, ek + 2:ek + 2 + Kc] == -1).all()
    # two "ranks" worth of rows back into contiguous arrays
    both = torch.cat([head, head])
    Bq = 2 * cap
    lab_o = torch.zeros(Bq, dtype=torch.int32, device="cuda")
    coef_o = torch.zeros(Bq, device="cuda")
    neg_o = torch.zeros(Bq, K, dtype=torch.int32, device="cuda")
    assert lib.tcar_shard_unpack_head(Bq, ek, K, p(both), ld, p(lab_o), p(coef_o), p(neg_o), None) == 0
    assert torch.equal(lab_o[:B], lab) and torch.equal(lab_o[cap:cap + B], lab) and (lab_o[B:cap] == -1).all()
    assert torch.equal(coef_o[:B], coef) and torch.equal(neg_o[cap:cap + B], neg) and (neg_o[B:cap] == -1).all()
    # packed item rows: [row | id | pad], shifted owner range, padding ids 0
    ldh, T, N = 256, 3, 500
    d = Dims(200, 250, 64, ldh, 64)                                       # a shard of 200 rows starting at id0 = 150
    nlive, ntot, ldr = B * T, cap * T, ldh + 4
    seq = torch.tensor(rng.randint(1, N + 1, nlive).astype(np.int32), device="cuda")
    rows = torch.zeros(ntot, ldr, device="cuda")
    rows[:, :ldh] = torch.tensor(rng.standard_normal((ntot, ldh)).astype(np.float32), device="cuda")
    ce, fb, loss = torch.rand(B, device="cuda"), torch.rand(B, device="cuda"), torch.zeros(B, device="cuda")
    assert lib.tcar_shard_pack_ids(nlive, ntot, ldh, p(seq), p(rows), ldr, B, p(ce), p(fb), 0.5, p(loss), None) == 0
    ri = rows.view(torch.int32)
    assert torch.equal(ri[:nlive, ldh], seq) and (ri[nlive:, ldh] == 0).all()
    np.testing.assert_allclose(loss.cpu().numpy(), (ce + 0.5 * fb).cpu().numpy(), rtol=1e-6)
    g = torch.zeros(200, ldh, device="cuda")
    assert lib.tcar_scatter_add_rows_packed(C.byref(d), p(rows), ldr, ntot, 150, p(g), None) == 0
    want = np.zeros((200, ldh), dtype=np.float64)
    sq, rw = seq.cpu().numpy(), rows[:nlive, :ldh].cpu().numpy().astype(np.float64)
    for r in range(nlive):
        if 150 < sq[r] <= 350:
            want[sq[r] - 151] += rw[r]
    np.testing.assert_allclose(g.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    # dense-weight norms: one workgroup per segment, fixed order -> bit-for-bit repeatable, equal to the fp64 sums
    segs = Segments()
    lens = [64, 250 * 256, 512 * 256, 4]
    arena = torch.tensor(rng.standard_normal(sum(lens)).astype(np.float32), device="cuda")
    segs.nseg, off = len(lens), 0
    for i, n in enumerate(lens):
        segs.off[i], segs.len[i], segs.slot[i] = off, n, i
        off += n
    outs = []
    for _ in range(3):
        sqn = torch.zeros(_lib.NSLOT, device="cuda")
        assert lib.tcar_sqnorm(p(arena), C.byref(segs), p(sqn), None) == 0
        outs.append(sqn.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    a64, off = arena.cpu().numpy().astype(np.float64), 0
    for i, n in enumerate(lens):
        np.testing.assert_allclose(outs[0][i], (a64[off:off + n] ** 2).sum(), rtol=1e-5)
        off += n
