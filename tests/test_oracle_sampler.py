"""The oracle sampler / metrics and the product host sampler / metrics against outputs of the REAL reference
(tests/golden/reference_*.json, produced by tests/golden/make_reference_fixtures.py)."""
import copy
import json
import os
import random

import numpy as np
import pytest

import tcar_amd  # noqa: F401
from oracle import metrics_oracle, sampler_oracle
from tcar_amd.host import metrics as host_metrics
from tcar_amd.host import sampler as host_sampler
from tcar_amd.host.data import bucketize_seconds
from helpers import GOLD, load_sampler_fixture


@pytest.fixture(scope="module")
def fx():
    return load_sampler_fixture()


def _norm(t):
    """6-tuple of nested lists / np ints -> plain python ints."""
    return json.loads(json.dumps(t, default=lambda o: int(o)))


def _run(cls, fx, split, seed, with_neg, **kw):
    len_d, sess, times = fx[split]
    random.seed(seed)
    np.random.seed(seed)
    extra = dict(neighbor_dict=fx["neighbor"], item_dict=fx["item_dict"], neg_num=fx["neg_num"]) if with_neg else {}
    s = cls(copy.deepcopy(len_d), sess, times, batch_size=16, **extra, **kw)
    out = []
    while s.has_next():
        out.append(_norm(s.next_batch()))
    return out


@pytest.mark.parametrize("cls", [sampler_oracle.OracleSampler, host_sampler.Sampler])
def test_train_batches_match_reference(fx, cls):
    got = _run(cls, fx, "train", fx["train_batches"]["seed"], True)
    want = fx["train_batches"]["batches"]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w


@pytest.mark.parametrize("cls", [sampler_oracle.OracleSampler, host_sampler.Sampler])
def test_test_batches_match_reference(fx, cls):
    got = _run(cls, fx, "test", fx["test_batches"]["seed"], False)
    want = fx["test_batches"]["batches"]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w


@pytest.mark.parametrize("cls", [sampler_oracle.OracleSampler, host_sampler.Sampler])
def test_bucket_of_exactly_batch_size_is_one_batch(fx, cls):
    e = fx["exact_bucket"]
    _, sess, times = fx["train"]
    random.seed(e["seed"])
    np.random.seed(e["seed"])
    s = cls({int(e["len"]): list(e["ids"])}, sess, times, None, fx["item_dict"], fx["neg_num"], batch_size=16)
    assert s.batch_num == e["batch_num"] == 1
    batches = s.batches if hasattr(s, "batches") else s.session_id_batches
    assert batches[0] == e["first_batch_ids"]


@pytest.mark.parametrize("cls", [sampler_oracle.OracleSampler, host_sampler.Sampler])
def test_alternative_negative_modes_match_reference(fx, cls):
    _, sess, times = fx["train"]
    e = fx["neg_neighbor"]
    random.seed(e["seed"])
    np.random.seed(e["seed"])
    s = cls({}, sess, times, fx["neighbor"], fx["item_dict"], fx["neg_num"], batch_size=16)
    for it, want in e["calls"]:
        assert [int(x) for x in s.neg_neighbor(it)] == want
    e = fx["neg_impression"]
    random.seed(e["seed"])
    np.random.seed(e["seed"])
    s = cls({}, sess, times, fx["impressions"], fx["item_dict"], fx["neg_num"], batch_size=16)
    for sid, want in e["calls"]:
        assert [int(x) for x in s.neg_neighbor_from_impre(sid)] == want


def test_bucketized_matches_reference(fx):
    secs = np.array(fx["bucketized"]["seconds"])
    want = np.array(fx["bucketized"]["bucket"])
    assert [int(sampler_oracle.bucketized(s)) for s in secs] == want.tolist()
    assert bucketize_seconds(secs).tolist() == want.tolist()
    assert want.max() == 11 and want[secs >= 1024].min() == 11      # the out-of-range id exists


def test_modes_agree_between_oracle_and_host(fx):
    """click_delta gaps and the per-session negative modes are commented-out call sites in the reference, so
    they cannot be pinned by running it; oracle (line-by-line restatement) and host (vectorised) must agree."""
    for kw in (dict(gap_mode="click_delta"), dict(neg_mode="neighbor")):
        a = _run(sampler_oracle.OracleSampler, fx, "train", 31, True, **kw)
        b = _run(host_sampler.Sampler, fx, "train", 31, True, **kw)
        assert a == b
    len_d, sess, times = fx["train"]
    outs = []
    for cls in (sampler_oracle.OracleSampler, host_sampler.Sampler):
        random.seed(8)
        np.random.seed(8)
        s = cls(copy.deepcopy(len_d), sess, times, fx["impressions"], fx["item_dict"], fx["neg_num"],
                batch_size=16, neg_mode="impression")
        o = []
        while s.has_next():
            o.append(_norm(s.next_batch()))
        outs.append(o)
    assert outs[0] == outs[1]


def test_cau_metrics_match_reference():
    with open(os.path.join(GOLD, "reference_metrics.json")) as fh:
        cases = json.load(fh)
    for c in cases:
        preds = np.array(c["preds"], dtype=np.float32)
        labels = np.array(c["labels"])
        for fn in (metrics_oracle.cau_metrics, host_metrics.cau_metrics):
            rec, mrr, ndcg = fn(preds, labels, c["cutoff"])
            assert [bool(x) for x in rec] == c["recall"]
            np.testing.assert_allclose(mrr, c["mrr"], rtol=1e-12)
            np.testing.assert_allclose(ndcg, c["ndcg"], rtol=1e-12)


def test_ild_unexp_host_vs_oracle():
    r = np.random.RandomState(1)
    n = 40
    cat = r.randint(0, 4, size=n)
    reverse_item = {i: 100 + i for i in range(n)}
    category_id = {100 + i: int(cat[i]) for i in range(n)}
    table = host_metrics.category_table(reverse_item, category_id, n)
    topk = np.stack([r.permutation(n)[:20] for _ in range(6)])
    seq = r.randint(1, n + 1, size=(6, 3))
    want_ild = [metrics_oracle.ild(t.tolist(), reverse_item, category_id) for t in topk]
    want_un = [metrics_oracle.unexp(s.tolist(), t.tolist(), reverse_item, category_id) for s, t in zip(seq, topk)]
    np.testing.assert_allclose(host_metrics.ild_batch(topk, table), want_ild, rtol=1e-12)
    np.testing.assert_allclose(host_metrics.unexp_batch(seq, topk, table), want_un, rtol=1e-12)
    # the product path: integer pair counts (device kernel tcar_eval_diversity; restated here in numpy) divided on the host
    c = table[topk]
    ild_cnt = (c[:, :, None] != c[:, None, :]).sum((1, 2))
    un_cnt = (c[:, :, None] != table[seq - 1][:, None, :]).sum((1, 2))
    ild, un = host_metrics.diversity_from_counts(ild_cnt, un_cnt, np.full(6, 20), 3)
    assert ild.tolist() == want_ild and un.tolist() == want_un          # exact: int / int in double precision, as Python
    # a list of 0 or 1 recommended items: getILD divides by n (n - 1) before getUnexp is reached (model_combine.py:182,304-305) —
    # the reference raises ZeroDivisionError, and so does the product path (no inf / nan leaks into the printed averages)
    for n_short in (0, 1):
        with pytest.raises(ZeroDivisionError):
            host_metrics.diversity_from_counts([0], [0], [n_short], 3)
        with pytest.raises(ZeroDivisionError):
            metrics_oracle.ild(list(range(n_short)), reverse_item, category_id)


@pytest.mark.parametrize("mode", ["neighbor", "impression"])
def test_fast_negative_modes_follow_the_reference_rules(fx, mode):
    """neg_fast=True does not replay the reference's random.choice order; it must still obey its rules: neighbour picks
    come from the label's list and differ from the label (sampler.py:133-140); impression picks come from the session's
    impression list mapped through item_dict, uniform padding only when fewer than K of 21 tries were catalog items
    (sampler.py:118-131)."""
    len_d, sess, times = fx["train"]
    src = fx["neighbor"] if mode == "neighbor" else fx["impressions"]
    random.seed(3)
    np.random.seed(3)
    s = host_sampler.Sampler(copy.deepcopy(len_d), sess, times, neighbor_dict=src, item_dict=fx["item_dict"],
                             neg_num=fx["neg_num"], batch_size=16, neg_mode=mode, neg_fast=True, verbose=False)
    seen = 0
    while s.has_next():
        keys = s.session_id_batches[s.batch_i]
        f = s.next_batch_arrays()
        neg, lab = f["neg"], f["label"]
        assert neg.shape == (len(keys), fx["neg_num"]) and neg.dtype == np.int32
        assert (neg >= 0).all() and (neg < len(fx["item_dict"])).all()
        for b, key in enumerate(keys):
            if mode == "neighbor":
                assert set(neg[b].tolist()) <= set(src[int(lab[b])]) and int(lab[b]) not in neg[b].tolist()
            else:
                mapped = {fx["item_dict"][x] - 1 for x in src[int(str(key).split('_')[0])] if x in fx["item_dict"]}
                frac_valid = len(mapped) / max(len(set(src[int(str(key).split('_')[0])])), 1)
                if frac_valid == 1.0:                      # every try is a catalog item: no uniform padding can occur
                    assert set(neg[b].tolist()) <= mapped
        seen += len(keys)
    assert seen > 0
