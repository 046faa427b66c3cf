"""Device-side batch formation and negative sampling (csrc/sampler.hip, tcar_amd.device_sampler) against the host sampler —
which reproduces the reference's Sampler bit for bit (tests/test_oracle_sampler.py) — and against the RULES of the
reference's negative modes (sampler.py:95-99,118-140)."""
import io
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

import tcar_amd  # noqa: F401

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _engine(fold, H=32, Ht=16):
    from tcar_amd.engine import TcarEngine
    from tcar_amd.host.model import initial_variables
    np.random.seed(1)
    params = initial_variables(fold.n_items, H, Ht, 0.3, 0.1, weight_seed=1)
    return TcarEngine(params, fold.content, fold.mwdhm, scoring="f32"), params


@pytest.mark.parametrize("gap_mode", ["active_t", "click_delta"])
def test_formed_batch_equals_the_host_batch(gap_mode):
    """every feed array of sampler.py:67-111 (inputs, label, publish fields, click week / hour of the LAST input click,
    dwell bucket in both gap modes), for several lengths incl. T = 1 and the longest bucket"""
    _need_gpu()
    from tcar_amd.device_sampler import DeviceSampler
    from tcar_amd.host.synth import SynthFold
    fold = SynthFold(n_items=700, dim=32, n_train=5000, n_test=10, seed=3, active_t=True)
    eng, _ = _engine(fold)
    st = fold.train
    ds = DeviceSampler(eng, st, "uniform")
    for T in (1, 2, 5, int(st.in_len.max())):
        idx = np.where(st.in_len == T)[0][:97]
        if len(idx) == 0:
            continue
        want = st.batch_arrays(idx, gap_mode)
        bt = ds.form(idx, 6, gap_mode)
        got = ds.read_back(bt)
        for k in ("seq", "pm", "pd", "pw", "ph", "pmi", "gap", "cw", "ch", "label"):
            assert np.array_equal(got[k], want[k]), (T, k)
        assert got["neg"].shape == (len(idx), 6) and got["neg"].min() >= 0 and got["neg"].max() < 700


def test_negative_modes_follow_the_reference_rules():
    _need_gpu()
    from tcar_amd.device_sampler import DeviceSampler
    from tcar_amd.host.synth import SynthFold
    N, K = 900, 20
    fold = SynthFold(n_items=N, dim=32, n_train=6000, n_test=10, seed=9)
    eng, _ = _engine(fold)
    st = fold.train
    idx = np.where(st.in_len == 2)[0][:512]
    lab = st.batch_arrays(idx, "active_t")["label"]
    # ---- uniform (sampler.py:98-99): K draws from [0, N) with replacement; the stream is keyed by (seed, counter, example)
    ds = DeviceSampler(eng, st, "uniform", seed=5)
    a = ds.read_back(ds.form(idx, K, counter=11))["neg"]
    b = ds.read_back(ds.form(idx, K, counter=11))["neg"]
    c = ds.read_back(ds.form(idx, K, counter=12))["neg"]
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.min() >= 0 and a.max() < N and abs(a.mean() - (N - 1) / 2) < 0.03 * N
    assert len(np.unique(a)) > 0.9 * N                                      # 10240 draws cover the catalog
    perm = np.random.RandomState(0).permutation(len(idx))
    d = ds.read_back(ds.form(idx[perm], K, counter=11))["neg"]
    assert np.array_equal(d, a[perm])                                       # keyed by the example, not by its batch row
    # ---- neighbour (sampler.py:133-140): picks from the label's list, never the label
    nb = fold.neighbor_dict(k=30)
    nb[int(lab[0])] = [int(x) for x in nb[int(lab[0])]] + [int(lab[0])] * 50   # a list polluted with its own key
    ds = DeviceSampler(eng, st, "neighbor", neighbor_dict=nb, seed=5)
    neg = ds.read_back(ds.form(idx, K))["neg"]
    assert (neg != lab[:, None]).all()
    for i in range(len(idx)):
        assert set(neg[i].tolist()) <= set(nb[int(lab[i])]), i
    assert len(set(neg[1].tolist())) > 5                                    # draws differ within a session
    # ---- impression (sampler.py:118-131): hits from the session's list (catalog items only) in front, uniform pads behind
    imp = fold.impression_dict(st)
    sids = st.impression_key[idx]
    all_valid, none_valid = int(sids[0]), int(sids[1])
    assert all_valid != none_valid
    imp[all_valid] = [10_000 + j for j in range(40, 70)]                    # every candidate is a catalog article
    imp[none_valid] = [90_000_000 + j for j in range(25)]                   # none is
    ds = DeviceSampler(eng, st, "impression", neighbor_dict=imp, item_dict=fold.item_dict, seed=5)
    neg = ds.read_back(ds.form(idx, K))["neg"]
    assert neg.min() >= 0 and neg.max() < N
    cat = {s: set(fold.item_dict[x] - 1 for x in imp[s] if x in fold.item_dict) for s in set(sids.tolist())}
    rows_all = np.where(sids == all_valid)[0]
    for i in rows_all:
        assert set(neg[i].tolist()) <= cat[all_valid]                       # 20 hits within <= 21 tries: all from the list
    # a list with no catalog article: every negative is a uniform pad -> they spread over the catalog
    pads = neg[np.where(sids == none_valid)[0]]
    assert pads.size and len(np.unique(pads)) > 0.6 * pads.size
    # mixed lists (~10 % unknown articles): the hits come first, and at most 21 tries were made
    for i in range(2, 60):
        s = int(sids[i])
        hits = [x in cat[s] for x in neg[i].tolist()]
        first_pad = hits.index(False) if False in hits else K
        assert first_pad >= 12, (i, first_pad)             # P(fewer than 12 hits in 21 tries at p = 0.9) < 1e-6


def test_training_with_the_device_sampler_matches_the_oracle_on_the_same_feed():
    """the step consumes the device-formed feed directly: read it back, give the SAME arrays to the oracle"""
    _need_gpu()
    from oracle.tcar_oracle import TcarOracle
    from tcar_amd.device_sampler import DeviceSampler
    from tcar_amd.host.synth import SynthFold
    fold = SynthFold(n_items=500, dim=32, n_train=3000, n_test=10, seed=4)
    eng, params = _engine(fold)
    ora = TcarOracle(params, fold.content, fold.mwdhm)
    st = fold.train
    ds = DeviceSampler(eng, st, "neighbor", neighbor_dict=fold.neighbor_dict(k=40), seed=3)
    for T in (2, 1, 3):
        idx = np.where(st.in_len == T)[0][:64]
        bt = ds.form(idx, 8, "click_delta")
        feed = ds.read_back(bt)
        got = eng.train_step(None, bt=bt).cpu().numpy()
        want = ora.train_step(feed).numpy()
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("chunk", [1, 2, 16])
def test_planned_schedule_forms_the_same_feeds_one_chunk_ahead(chunk):
    """DeviceSampler.plan + planned (the trainer loop's path: indices resident, the feeds of chunk j + 1 formed on a side stream
    while the steps of chunk j run, two alternating chunk buffers, one event pair per chunk) yields, batch for batch, the feed
    that form() builds for the same examples and counter — while training steps consume them.  chunk = 1: a hand-off per batch
    (round 3's form), 2: the buffers alternate within the schedule, 16: one chunk."""
    _need_gpu()
    from tcar_amd.device_sampler import DeviceSampler
    from tcar_amd.host.synth import SynthFold
    fold = SynthFold(n_items=500, dim=32, n_train=3000, n_test=10, seed=4)
    eng, _ = _engine(fold)
    st = fold.train
    sched = []
    for T in (2, 1, 3, 2, 1):
        ids = np.where(st.in_len == T)[0]
        sched.append(ids[len(sched) * 7:len(sched) * 7 + (64 if T != 3 else 33)])
    ref = DeviceSampler(eng, st, "uniform", seed=11)
    want = [ref.read_back(ref.form(ids, 6, "click_delta", counter=i)) for i, ids in enumerate(sched)]
    ds = DeviceSampler(eng, st, "uniform", seed=11)
    ds.CHUNK = chunk
    ds.plan(sched)
    got = []
    for bt in ds.planned(6, "click_delta"):
        eng.train_step(None, bt=bt, defer_update=True)          # the consumer's step reads the feed while the next one is formed
        n = 7 * bt.B * bt.T + 3 * bt.B + bt.B * bt.K
        got.append((bt.B, bt.T, bt._keep[:n].clone()))
    eng.flush()
    torch.cuda.synchronize()
    assert len(got) == len(sched)
    for (B, T, feed), w, ids in zip(got, want, sched):
        assert B == len(ids) and T == w["seq"].shape[1]
        f = feed.cpu().numpy()
        n = B * T
        assert np.array_equal(f[:n].reshape(B, T), w["seq"])
        assert np.array_equal(f[7 * n + 2 * B:7 * n + 3 * B], w["label"])
        assert np.array_equal(f[7 * n + 3 * B:].reshape(B, 6), w["neg"])
        assert np.array_equal(f[6 * n:7 * n].reshape(B, T), w["gap"])


def test_cli_trains_with_the_device_sampler():
    _need_gpu()
    from tcar_amd.host.cli import main
    common = ["--synthetic", "600", "--synthetic_train", "4000", "--synthetic_test", "300", "--hidden_size", "48",
              "--time_hidden_size", "16", "--batch_size", "128", "--gap_mode", "click_delta", "--epoch", "2"]
    res = {}
    for mode in ("uniform", "neighbor", "impression"):
        buf = io.StringIO()
        with redirect_stdout(buf):
            m = main(common + ["--neg_mode", mode, "--device_sampler", "1"])
        out = buf.getvalue()
        assert "Recall@20" in out and np.isfinite(m.last_metrics["loss"])
        losses = [float(l.split("loss:")[1]) for l in out.splitlines() if l.startswith("\tloss:")]
        assert len(losses) == 2 and losses[1] < losses[0], (mode, losses)
        res[mode] = m.last_metrics["recall"]
    # same data, same shuffles: the host-sampled run lands in the same place (different negative streams)
    buf = io.StringIO()
    with redirect_stdout(buf):
        m = main(common + ["--neg_mode", "uniform"])
    assert abs(m.last_metrics["recall"] - res["uniform"]) < 0.05
