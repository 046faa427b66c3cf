"""CPU restatement of the reference evaluation metrics.  TEST INFRASTRUCTURE ONLY.

  cau_metrics   util.py:8-18
  top-k list    model_combine.py:301   np.argsort(pred).tolist()[::-1][:20]
  getILD        model_combine.py:174-182
  getUnexp      model_combine.py:184-194

PARITY STATUS: ``cau_metrics`` is pinned by ``tests/golden/reference_metrics.json``
(outputs of the real ``util.cau_metrics``).  ILD / unexp are methods of the TF model
class and cannot be imported; they are restated from the source lines above.
"""
import numpy as np


def cau_metrics(preds, labels, cutoff=20):
    """rank = 1 + #{j : preds[j] > preds[label]} (strict: ties do not count against the label)."""
    recall, mrr, ndcg = [], [], []
    for row, lab in zip(preds, labels):
        rank = int((row[lab] < row).sum()) + 1
        hit = rank <= cutoff
        recall.append(hit)
        mrr.append(1.0 / rank if hit else 0.0)
        ndcg.append(1.0 / np.log2(rank + 1) if hit else 0.0)
    return recall, mrr, ndcg


def topk_list(pred, k=20):
    """model_combine.py:301: reversed ascending argsort => among equal scores the HIGHER index comes first
    (numpy's default quicksort is not stable in general; for the tie-free fp32 rows this is exact)."""
    return np.argsort(pred, kind="stable").tolist()[::-1][:k]


def ild(rec, reverse_item, category_id):
    """fraction of ordered pairs (i != j) of the list with different category."""
    n = len(rec)
    s = 0
    for i in range(n):
        for j in range(n):
            if j != i and category_id[reverse_item[rec[i]]] != category_id[reverse_item[rec[j]]]:
                s += 1
    return s / (n * (n - 1))


def unexp(in_seq, rec, reverse_item, category_id):
    """fraction of (recommended, input) pairs with different category; input ids are 1-based."""
    n = len(rec)
    if n == 0:
        return 0
    s = 0
    for i in range(n):
        for ini in in_seq:
            if category_id[reverse_item[rec[i]]] != category_id[reverse_item[ini - 1]]:
                s += 1
    return s / (n * len(in_seq))
