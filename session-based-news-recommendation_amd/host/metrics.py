"""Ranking / diversity metrics of the evaluation loop (util.py:8-18, model_combine.py:174-194,301-313).

The GPU path returns, per session, the label's rank and the top-k indices (kernel `tcar_rank_topk`), so the
host never sees the [B,N] score matrix; the functions here turn ranks / top-k lists into the numbers the
reference prints.  `cau_metrics` keeps the reference signature for callers that do hold score rows.
"""
from __future__ import annotations

import numpy as np


def metrics_from_ranks(ranks: np.ndarray, cutoff: int = 20):
    """ranks are 1-based.  Returns (hit, mrr, ndcg) arrays, element-wise as util.py:15-17."""
    ranks = np.asarray(ranks, dtype=np.int64)
    hit = ranks <= cutoff
    mrr = np.where(hit, 1.0 / ranks, 0.0)
    ndcg = np.where(hit, 1.0 / np.log2(ranks + 1.0), 0.0)
    return hit, mrr, ndcg


def cau_metrics(preds, labels, cutoff=20):
    """Same contract as util.py:8-18: rank = 1 + #{j: preds[j] > preds[label]} (strict)."""
    preds = np.asarray(preds)
    labels = np.asarray(labels, dtype=np.int64)
    if preds.ndim == 1:
        preds = preds[None, :]
    lab = preds[np.arange(len(labels)), labels]
    ranks = (preds > lab[:, None]).sum(1) + 1
    hit, mrr, ndcg = metrics_from_ranks(ranks, cutoff)
    return hit.tolist(), mrr.tolist(), ndcg.tolist()


def ild_batch(topk: np.ndarray, cat_of_item: np.ndarray) -> np.ndarray:
    """model_combine.py:174-182 for a [B,k] array of 0-based item ids: share of ordered pairs (i != j) whose
    categories differ."""
    c = cat_of_item[topk]                                     # [B,k]
    k = c.shape[1]
    diff = (c[:, :, None] != c[:, None, :]).sum((1, 2))       # the diagonal never differs
    return diff / float(k * (k - 1))


def unexp_batch(seq: np.ndarray, topk: np.ndarray, cat_of_item: np.ndarray) -> np.ndarray:
    """model_combine.py:184-194: share of (recommended, input) pairs with different category; `seq` holds
    1-based ids."""
    cr = cat_of_item[topk]                                    # [B,k]
    ci = cat_of_item[np.asarray(seq) - 1]                     # [B,T]
    diff = (cr[:, :, None] != ci[:, None, :]).sum((1, 2))
    return diff / float(cr.shape[1] * ci.shape[1])


def diversity_from_counts(ild_cnt, unexp_cnt, n_rec, T: int):
    """The divisions of getILD / getUnexp (model_combine.py:182,194) on the integer pair counts the device kernel
    tcar_eval_diversity returns: score / (n (n - 1)) and score / (n len(inSeq)) — Python int / int, i.e. double precision."""
    ild_cnt = np.asarray(ild_cnt, dtype=np.int64)
    unexp_cnt = np.asarray(unexp_cnt, dtype=np.int64)
    n = np.asarray(n_rec, dtype=np.int64)
    if n.size and int(n.min()) <= 1:
        # getILD divides by n (n - 1) unguarded (model_combine.py:182): a recommendation list of 0 or 1 items raises there, and here
        raise ZeroDivisionError("getILD: a recommendation list of %d item(s) (model_combine.py:182 divides by n (n - 1))" % int(n.min()))
    ild = ild_cnt / (n * (n - 1)).astype(np.float64)
    unexp = np.where(n > 0, unexp_cnt / np.maximum(n * int(T), 1).astype(np.float64), 0.0)      # getUnexp returns 0 for n == 0
    return ild, unexp


def category_table(reverse_item: dict, category_id, n_items: int) -> np.ndarray:
    """int table cat[item0] = code of category_id[reverse_item[item0]] (the double lookup of model_combine.py:180).
    The reference only ever compares categories with `!=`, so any label type works (MIND's categories are strings): the
    labels are factorised to integer codes.  Items without a category (the reference raises KeyError lazily, when such an
    item is first recommended) get one code of their own each, i.e. they differ from everything else."""
    labels, missing = [], []
    for i in range(n_items):
        orig = reverse_item.get(i) if hasattr(reverse_item, "get") else reverse_item[i]
        if orig is not None and orig in category_id:
            labels.append(category_id[orig])
        else:
            labels.append(None)
            missing.append(i)
    if missing:
        # the reference would raise KeyError the first time one of these is recommended (model_combine.py:180): say once that
        # ILD / unexp are computed on a fold where the reference's evaluation could fail
        import sys
        print("[tcar] %d of %d catalog items have no category: each counts as a category of its own in ILD / unexp "
              "(the reference raises KeyError when one is recommended)" % (len(missing), n_items), file=sys.stderr)
    out = np.empty(n_items, dtype=np.int64)
    known = [l for l in labels if l is not None]
    codes = {}
    for l in known:
        if l not in codes:
            codes[l] = len(codes)
    nxt = len(codes)
    for i, l in enumerate(labels):
        if l is None:
            out[i] = nxt
            nxt += 1
        else:
            out[i] = codes[l]
    return out
