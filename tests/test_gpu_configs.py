"""BASELINE.json configurations 3-5 on the GPU.

* Adressa-like fold (active_t dwell seconds incl. the out-of-range bucket 11, impression negatives, sampler.py:96,118-131)
  and MIND-like fold (one click time per session, active_t = 1, neighbour negatives of generate_neighbor.py,
  sampler.py:97,133-140): engine-vs-oracle step parity at the configurations' catalog sizes with negatives drawn by the
  EXACT-replay sampler modes (reference-pinned on the CPU by tests/test_oracle_sampler.py), so both sides see identical ids.
* Synthetic 10M-item catalog, d = 256: size-independent properties (no oracle run at this size).
* Same step twice: which variables repeat bit for bit.
"""
import random

import numpy as np
import pytest
import torch

import tcar_amd  # noqa: F401

pytestmark = pytest.mark.gpu
RTOL = 1e-3            # north-star tolerance (BASELINE.json): 1e-3 relative


def close(got, want, rtol=RTOL, atol_scale=2e-5, name=""):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    atol = atol_scale * max(1e-30, float(np.abs(want).max()))
    bad = np.abs(got - want) > atol + rtol * np.abs(want)
    assert not bad.any(), "%s: %d / %d off, max abs err %.3e (max |want| %.3e)" % (
        name, int(bad.sum()), bad.size, float(np.abs(got - want).max()), float(np.abs(want).max()))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


CONFIGS = {
    # name: (n_items, fold kwargs, neg_mode)
    "adressa_impression": (15000, dict(active_t=True), "impression"),
    "mind_neighbor": (30000, dict(active_t=True, same_click_time=True), "neighbor"),
}


@pytest.mark.parametrize("scoring", ["f32", "bf16x3", "bf16x3-mixed"])
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_step_parity_with_reference_negative_modes(name, scoring):
    _need_gpu()
    from oracle.tcar_oracle import TcarOracle
    from tcar_amd.engine import TcarEngine
    from tcar_amd.host.model import initial_variables
    from tcar_amd.host.sampler import Sampler
    from tcar_amd.host.synth import SynthFold
    n_items, fkw, neg_mode = CONFIGS[name]
    H, Ht, B, K = 250, 64, 96, 20
    fold = SynthFold(n_items=n_items, dim=H, n_train=6000, n_test=300, seed=23, **fkw)
    st = fold.train
    if name.startswith("adressa"):
        # some dwell times beyond 1023 s: bucket 11 indexes past the 11-row duration table (sampler.py:18-21, DESIGN S7)
        st.gap_active[::37] = 11
    else:
        assert (st.gap_active == 1).all() and (st.gap_delta == 0).all()      # mind_preprocess.py:22; shared click time
    src = fold.impression_dict(st) if neg_mode == "impression" else fold.neighbor_dict()
    len_dict = {int(T): np.where(st.in_len == T)[0].tolist() for T in np.unique(st.in_len)}
    random.seed(7)
    np.random.seed(7)
    smp = Sampler(len_dict, None, None, src, fold.item_dict, K, batch_size=B, gap_mode="active_t", neg_mode=neg_mode,
                  store=st, verbose=False, neg_fast=False)                   # exact replay of sampler.py:118-140
    batches = []
    while smp.has_next() and len(batches) < 4:
        b = smp.next_batch_arrays()
        if b["seq"].shape[0] >= 24:
            batches.append(b)
    assert len(batches) == 4
    for b in batches:
        assert b["neg"].shape == (b["seq"].shape[0], K)
        if neg_mode == "neighbor":                                           # picks differ from the label (sampler.py:137)
            assert (b["neg"] != b["label"][:, None]).all()
            assert all(set(b["neg"][i].tolist()) <= set(src[int(b["label"][i])]) for i in range(len(b["label"])))
    np.random.seed(11)
    params = initial_variables(n_items, H, Ht, 0.25, 0.1, weight_seed=5)      # norms > 1 on many rows: the clip is active
    eng = TcarEngine(params, fold.content, fold.mwdhm, lr=2e-3, scoring=scoring)
    ora = TcarOracle(params, fold.content, fold.mwdhm, lr=2e-3)
    # evaluation of a test batch at the initial variables: per-session logits, CE and ranks (north star: 1e-3 relative)
    te = fold.test
    idx = np.where(te.in_len == 2)[0][:64]
    tb = te.batch_arrays(idx, "active_t")

    def check_eval(tight):
        rank, topk, ce, logits = eng.eval_step(tb, keep_logits=True)
        lo, ce_o = ora.eval_batch(tb)
        if tight:
            close(logits.cpu().numpy(), lo.numpy(), name="logits", atol_scale=1e-4)
            close(ce.cpu().numpy(), ce_o.numpy(), name="ce")
        else:
            # after Adam steps a few ill-conditioned coordinates have moved by +-lr per step on either side (see below):
            # the scores agree norm-wise and the batch loss at the gate
            d = logits.cpu().numpy().astype(np.float64) - lo.numpy()
            # (mixed: four Adam steps on gradients with 1e-2 bf16 noise move more coordinates the other way)
            loose = scoring == "bf16x3-mixed"
            assert np.sqrt((d * d).mean()) <= (1e-2 if loose else 2e-3) * np.sqrt((lo.numpy() ** 2).mean()), "logits after training"
            assert abs(float(ce.mean()) - float(ce_o.mean())) <= (3e-3 if loose else 1e-3) * float(ce_o.mean())
        lab = torch.as_tensor(tb["label"], dtype=torch.long)
        want_rank = ((lo > lo.gather(1, lab[:, None])).sum(1) + 1).numpy()
        r = rank.cpu().numpy()
        # ranks can differ where two scores are within rounding of each other: allow near ties, nothing else
        off = np.nonzero(r != want_rank)[0]
        tol = 1e-3 if tight else 5e-3
        for i in off:
            row = lo[i].numpy()
            gap = np.abs(row - row[tb["label"][i]])
            assert abs(int(r[i]) - int(want_rank[i])) <= int((gap < tol * max(1e-6, np.abs(row).max())).sum()), (i, r[i], want_rank[i])
        if tight:
            assert len(off) <= max(3, len(r) // 8)
        hr_e, hr_o = float((r <= 20).mean()), float((want_rank <= 20).mean())
        assert abs(hr_e - hr_o) <= 0.002 + 1.0 / len(r)

    check_eval(tight=True)
    # first batch: loss, every gradient, clip norms
    loss = eng.loss_and_grads(batches[0])
    o, g_o, sq_o = ora.loss_and_grads(batches[0])
    close(loss.cpu().numpy(), o["loss"].detach().numpy(), name="loss")
    g_e, sq_e = eng.export_grads(), eng.export_sqnorms()
    # A gradient that is zero in exact arithmetic comes out as fp32 rounding noise on both sides (MIND: every dwell id is 1,
    # so the dwell projection shifts all attention scores of a session alike): the absolute floor is 1e-7 of the largest
    # gradient entry of the step, far below anything the update can see.
    gmax = max(float(np.abs(v.numpy()).max()) for v in g_o.values())
    mixed = scoring == "bf16x3-mixed"      # gradient GEMMs on plain bf16 operands: every gradient 1e-2 norm-wise, norms 2e-2
    for k in g_o:
        want = g_o[k].numpy()
        if mixed:
            err = float(np.linalg.norm(np.asarray(g_e[k], dtype=np.float64) - want))
            assert err <= 1e-2 * np.linalg.norm(want) + 1e-7 * gmax * np.sqrt(want.size), ("grad " + k, err, float(np.linalg.norm(want)))
            assert abs(sq_e[k] - sq_o[k]) <= 2e-2 * sq_o[k] + 1e-12 * gmax * gmax * want.size, ("sqnorm", k, sq_e[k], sq_o[k])
            continue
        scale = max(5e-5, 1e-7 * gmax / max(1e-30, float(np.abs(want).max())))
        close(g_e[k], want, name="grad " + k, atol_scale=scale)
        assert abs(sq_e[k] - sq_o[k]) <= 2e-3 * sq_o[k] + 1e-12 * gmax * gmax * want.size, ("sqnorm", k, sq_e[k], sq_o[k])
    # training steps over all four batches
    for i, b in enumerate(batches):
        le, lo = eng.train_step(b).cpu().numpy(), ora.train_step(b).numpy()
        if mixed and i:
            # after Adam steps on bf16-noisy gradients: 1e-2 per session; and the negative term is DISCONTINUOUS in the
            # reference's own fp32 (S8: sigmoid rounds to 1 near x = 16.6, the term jumps to 55.26): a session whose negative
            # logit sits at that edge may land on the other side (0.01 * ~38 = 0.39 of loss) — at most 2 % of the sessions
            bad = np.abs(le - lo) > 1e-2 * np.abs(lo) + 2e-5 * np.abs(lo).max()
            assert bad.sum() <= max(1, 0.02 * bad.size), ("train loss", int(bad.sum()), bad.size)
            assert np.abs(le - lo)[bad].max(initial=0.0) <= 0.01 * 56.0, float(np.abs(le - lo).max())
        else:
            close(le, lo, name="train loss")
    # Variables after the four Adam steps.  Adam normalises by sqrt(v): a coordinate whose gradient sits at rounding level
    # moves by ~lr per step with a rounding-determined sign, so a handful of coordinates may differ by up to 2 * lr * steps
    # while everything else holds the relative gate.
    p_e, p_o = eng.export_params(), ora.export()
    lr, steps = 2e-3, len(batches)
    for k in p_o:
        ge, go = np.asarray(p_e[k], dtype=np.float64), np.asarray(p_o[k], dtype=np.float64)
        err = np.abs(ge - go)
        bad = err > 1e-4 * max(1e-30, np.abs(go).max()) + RTOL * np.abs(go)
        if not mixed and (k == "item_emb" or k.startswith("attout_")):      # well-conditioned gradients (the scoring side)
            assert bad.sum() <= max(8, 2e-3 * bad.size), ("param", k, int(bad.sum()), bad.size)
        assert err.max() <= 2.2 * lr * steps, ("param", k, float(err.max()))
    check_eval(tight=False)


def _host_mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def test_stress_10m_items_d256_properties():
    """BASELINE configs[4]: N = 10,000,000 items, d = 256 for item and content (K_score = 256 + 256 + 320), B = 512 on ONE
    GPU (~150 GB of HBM).  rank / CE / top-k row checks against the materialised scores, the last row against a direct fp64
    dot product, the loss falls over training steps, padding stays zero."""
    _need_gpu()
    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs ~150 GB of device memory")
    if _host_mem_available_gb() < 60:
        pytest.skip("needs ~25 GB of host memory for the tables")
    from tcar_amd.engine import TcarEngine
    from tcar_amd.host.model import initial_variables
    N, H, Ht, B, T, K = 10_000_000, 256, 64, 512, 2, 20
    rng = np.random.RandomState(7)
    np.random.seed(7)
    params = initial_variables(N, H, Ht, 0.05, 0.05, weight_seed=7, lean=True)
    g32 = np.random.default_rng(3)
    content = np.empty((N + 1, H), dtype=np.float32)
    for lo in range(0, N + 1, 1 << 20):
        hi = min(N + 1, lo + (1 << 20))
        content[lo:hi] = g32.standard_normal((hi - lo, H), dtype=np.float32) * np.float32(0.5)
    content[0] = 0
    mw = np.stack([rng.randint(1, 13, N), rng.randint(1, 32, N), rng.randint(1, 8, N), rng.randint(1, 25, N),
                   rng.randint(1, 61, N)], -1).astype(np.int32)
    b = {"seq": rng.randint(1, N + 1, (B, T)), "label": rng.randint(0, N, B), "pm": rng.randint(1, 13, (B, T)),
         "pd": rng.randint(1, 32, (B, T)), "pw": rng.randint(1, 8, (B, T)), "ph": rng.randint(1, 25, (B, T)),
         "pmi": rng.randint(1, 61, (B, T)), "cw": rng.randint(0, 7, B), "ch": rng.randint(0, 24, B),
         "gap": rng.randint(0, 11, (B, T)), "neg": rng.randint(0, N, (B, K))}
    b = {k: v.astype(np.int32) for k, v in b.items()}
    b["label"][-1] = N - 1                    # the very last score of the last row
    b["seq"][-1, :] = N                       # highest item row on the session side
    b["neg"][-1, :] = N - 1
    eng = TcarEngine(params, content, mw, scoring="bf16x3")
    rank, topk, ce, logits = eng.eval_step(b, keep_logits=True)
    lab = torch.as_tensor(b["label"], dtype=torch.long, device="cuda")
    for r in (0, B // 2, B - 1):                                       # row-wise, to keep the fp64 copies small
        lg = logits[r].double()
        assert int(rank[r]) == int((lg > lg[lab[r]]).sum()) + 1, r
        assert abs(float(ce[r]) - float(torch.logsumexp(lg, 0) - lg[lab[r]])) < 1e-3, r
        tv = lg[topk[r].long()]
        assert (tv[:-1] >= tv[1:]).all() and float(tv[-1]) >= float(lg.topk(21).values[20])
    # scores of the last session against fp64 dot products of the fp32 operands (item | content columns + time columns)
    att = eng.attout[B - 1].double()
    g = eng.geo
    for n in (0, N // 2, N - 1):
        row = eng.E[n].double()                                          # item | content (the fp32 time block is not kept
        want = float((row[:g.ic] * att[:g.ic]).sum())                    # in the bf16 modes: rebuild it from the tables)
        for k, name in enumerate(["month", "day", "week", "hour", "minute"]):
            sg = eng.seg[name]
            tab = eng.W[sg["off"]:sg["off"] + sg["n"]].view(sg["rows"], sg["cols"])[int(mw[n, k])].double()
            tab = tab / max(1.0, float(tab.norm()))
            want += float((tab * att[g.ic + k * g.ldt:g.ic + (k + 1) * g.ldt]).sum())
        assert abs(float(logits[B - 1, n]) - want) <= 1e-3 * abs(want) + 1e-4, (n, float(logits[B - 1, n]), want)
    del logits
    first = eng.train_step(b).clone()
    l0 = float(first.sum())
    for _ in range(3):
        l1 = float(eng.train_step(b).sum())
    assert np.isfinite(l1) and l1 < l0
    if g.Npad > N:
        assert float(eng.E[N:].abs().max()) == 0.0                       # padding rows untouched
    if g.ldh > H:
        assert float(eng.E[:, H:g.ldh].abs().max()) == 0.0
    first = first.cpu().numpy()
    # ---- the same configuration in the BENCHED precision (bf16x3-mixed: softmax epilogue, one-hot forms, hi-only gradient GEMMs;
    # VERDICT r04 item 7).  Its forward is the two-plane forward: the per-session loss of the first step (same variables, same
    # batch) agrees with the materialised-logits engine's at the 1e-3 gate; training moves the loss down; padding stays zero.
    del eng, rank, topk, ce
    torch.cuda.empty_cache()
    eng = TcarEngine(params, content, mw, scoring="bf16x3-mixed")
    del params
    form = eng.step_form(eng.make_resident(b))
    assert form["fused_ce"], form                                        # no [B, N] fp32 logits in this step
    m_first = eng.train_step(b).clone().cpu().numpy()
    np.testing.assert_allclose(m_first, first, rtol=1e-3, atol=1e-5)
    m0 = float(m_first.sum())
    for _ in range(3):
        m1 = float(eng.train_step(b).sum())
    assert np.isfinite(m1) and m1 < m0
    eng.check_forks()
    if eng.geo.Npad > N:
        assert float(eng.E[N:].abs().max()) == 0.0


@pytest.mark.parametrize("scoring,T", [("f32", 3), ("bf16x3", 3), ("bf16x3", 5), ("bf16x3-mixed", 7)])
def test_same_step_twice_bitwise_report(scoring, T):
    """SURVEY.md §5 'race detection' row: the same fused training step from the same state, three times.  Every quantity that
    is accumulated in a fixed order must repeat bit for bit — engine.DETERMINISTIC_GRADS names them (in the split-bf16 modes:
    every gradient, every clip norm, every variable after one AND after two steps); what still goes through float atomics
    (fp32 mode: biases and residual weights) is listed and must agree to rounding.  The set of non-repeating variables may
    only shrink."""
    _need_gpu()
    from tcar_amd.engine import DETERMINISTIC_GRADS, TcarEngine, VAR_ORDER
    from tcar_amd.host.model import initial_variables
    from tcar_amd.host.synth import SynthFold
    N, H, Ht, B, K = 5000, 250, 64, 512, 20
    fold = SynthFold(n_items=N, dim=H, n_train=80000, n_test=10, seed=5)     # Zipf items: popular rows repeat in a batch
    idx = np.where(fold.train.in_len == T)[0][:B]        # T = 5, 7: more than 1,536 batch rows -> the weight gradients split K
    assert len(idx) == B
    batch = fold.train.batch_arrays(idx, "click_delta")
    batch["neg"] = np.random.RandomState(2).randint(0, N, size=(len(idx), K)).astype(np.int32)
    batch["neg"][:, 0] = batch["neg"][0, 0]                                  # one negative row with 512 sources
    assert len(np.unique(batch["seq"])) < batch["seq"].size                  # repeated ids: the scatter has collisions
    np.random.seed(4)
    params = initial_variables(N, H, Ht, 0.25, 0.1, weight_seed=4)
    runs = []
    for _ in range(3):
        eng = TcarEngine(params, fold.content, fold.mwdhm, scoring=scoring)
        eng.train_step(batch)                       # the fused step: forward, backward, clip + Adam
        g = eng.export_grads()
        sq = eng.export_sqnorms()
        p1 = eng.export_params()
        eng.train_step(batch)
        p2 = eng.export_params()
        runs.append((g, p1, p2, sq))
        del eng
    differ = {"grad": [], "param": [], "param after 2 steps": []}
    for k in VAR_ORDER:
        for i, tag in enumerate(differ):
            same = all(np.array_equal(runs[0][i][k], r[i][k]) for r in runs[1:])
            if not same:
                differ[tag].append(k)
                for r in runs[1:]:
                    close(r[i][k], runs[0][i][k], rtol=1e-4, atol_scale=2e-5 if i == 0 else 2e-4, name="%s %s repeat" % (tag, k))
    print("not bitwise repeatable:", differ)
    # split-bf16 modes: EVERYTHING repeats, also after the second step; fp32 mode keeps float atomics for the biases and the
    # residual weights (the order-fixed column sums belong to the fused query chain of the split-bf16 step)
    from tcar_amd.engine import ATOMIC_IN_F32
    assert set(DETERMINISTIC_GRADS) == set(VAR_ORDER) and len(ATOMIC_IN_F32) == 6
    must = DETERMINISTIC_GRADS if scoring != "f32" else tuple(k for k in VAR_ORDER if k not in ATOMIC_IN_F32)
    for k in must:
        assert k not in differ["grad"] and k not in differ["param"], ("lost determinism", k, differ)
        assert all(r[3][k] == runs[0][3][k] for r in runs[1:]), ("norm of %s not repeatable" % k, [r[3][k] for r in runs])
    assert set(differ["grad"]) <= set(VAR_ORDER) - set(must)
    if scoring != "f32":
        assert not differ["param after 2 steps"], differ
    if T > 3:
        # the K splits of the weight gradients are slabs folded in split order: same values as the un-split launch, to rounding
        eng = TcarEngine(params, fold.content, fold.mwdhm, scoring=scoring)
        eng.set_tuning(TCAR_WGRAD_KS=1 << 20)          # this engine's own copy of the switches
        eng.train_step(batch)
        g1 = eng.export_grads()
        for k in VAR_ORDER:
            if k.endswith("/w_3d") or k.endswith("/w1"):
                close(runs[0][0][k], g1[k], rtol=1e-4, atol_scale=1e-5, name="split vs un-split " + k)
