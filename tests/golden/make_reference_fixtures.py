"""Generate golden vectors by RUNNING the real reference code that is importable in the build container.

Only works where /root/reference exists (the build container).  It imports
  /root/reference/sampler.py   (numpy + random only)
  /root/reference/util.py      (needs a stub `tensorflow` module; only cau_metrics is used)
feeds them a tiny synthetic dataset and writes inputs + outputs as JSON next to this script:
  reference_sampler.json   Sampler batches (train mode with uniform negatives, test mode), direct calls of
                           neg_neighbor / neg_neighbor_from_impre, bucketized(0..2000)
  reference_metrics.json   cau_metrics on random rows and tie cases
  reference_neighbor.json  data_process/generate_neighbor.get_neighbor on three publish-time vectors (300 ints with ties,
                           130 datetimes with duplicates, 60 ints: the pad-to-100 branch)
No reference source is copied; the JSON holds data only.   Usage:  python tests/golden/make_reference_fixtures.py
"""
import copy
import datetime
import json
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def _import_reference():
    sys.modules.setdefault("tensorflow", types.ModuleType("tensorflow"))
    sys.path.insert(0, REF)
    import sampler as ref_sampler   # noqa
    import util as ref_util         # noqa
    sys.path.remove(REF)
    return ref_sampler, ref_util


def _jsonable(o):
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, (np.bool_,)):
        return bool(o)
    if isinstance(o, np.ndarray):
        return o.tolist()
    if isinstance(o, (datetime.datetime,)):
        return o.isoformat()
    if isinstance(o, (list, tuple)):
        return [_jsonable(x) for x in o]
    if isinstance(o, dict):
        return {str(k): _jsonable(v) for k, v in o.items()}
    return o


def main():
    ref_sampler, ref_util = _import_reference()
    import tcar_amd  # noqa: F401
    from tcar_amd.host.synth import SynthFold

    fold = SynthFold(n_items=60, dim=8, n_train=150, n_test=40, seed=7, active_t=True)
    len_d, sess_d, time_d = fold.to_dicts(fold.train, with_active=True)
    tlen_d, tsess_d, ttime_d = fold.to_dicts(fold.test, with_active=True)
    item_dict = fold.item_dict
    neighbor = fold.neighbor_dict(k=5)                                   # 0-based item -> 0-based neighbours
    rs = np.random.RandomState(3)
    impressions = {}                                                     # int session id -> original article ids
    for key in sess_d:
        sid = int(key.split("_")[0])
        if sid not in impressions:
            cand = [int(10_000 + x) for x in rs.randint(0, 60, size=rs.randint(2, 9))]
            cand += [int(99_000 + x) for x in rs.randint(0, 5, size=2)]  # ids that are NOT in item_dict
            impressions[sid] = cand

    out = {"dataset": {
        "train": {"len_dict": len_d, "session_dict": sess_d, "session_time_dict": time_d},
        "test": {"len_dict": tlen_d, "session_dict": tsess_d, "session_time_dict": ttime_d},
        "item_dict": item_dict, "neighbor": neighbor, "impressions": impressions, "neg_num": 6}}

    # (1) train-mode sampler, as shipped: active_t gaps + uniform negatives (sampler.py:87,98-99)
    random.seed(2020)
    np.random.seed(2020)
    s = ref_sampler.Sampler(copy.deepcopy(len_d), sess_d, time_d, neighbor, item_dict, 6, batch_size=16)
    batches = []
    while s.has_next():
        batches.append(s.next_batch())
    out["train_batches"] = {"seed": 2020, "batch_size": 16, "batches": batches}

    # (2) test-mode sampler: no neighbour dict -> neg = [] (model_combine.py:261)
    random.seed(11)
    np.random.seed(11)
    s = ref_sampler.Sampler(copy.deepcopy(tlen_d), tsess_d, ttime_d, batch_size=16)
    batches = []
    while s.has_next():
        batches.append(s.next_batch())
    out["test_batches"] = {"seed": 11, "batch_size": 16, "batches": batches}

    # (3) a bucket of exactly batch_size sessions is ONE batch (sampler.py:42 uses '>')
    random.seed(5)
    np.random.seed(5)
    one_len = max(len_d, key=lambda k: len(len_d[k]))
    exact = {one_len: list(len_d[one_len])[:16]}
    s = ref_sampler.Sampler(copy.deepcopy(exact), sess_d, time_d, None, item_dict, 6, batch_size=16)
    out["exact_bucket"] = {"seed": 5, "len": one_len, "ids": exact[one_len], "batch_num": s.batch_num,
                           "first_batch_ids": s.session_id_batches[0]}

    # (4) the two alternative negative modes, called directly (sampler.py:118-140)
    random.seed(99)
    np.random.seed(99)
    s = ref_sampler.Sampler({}, sess_d, time_d, neighbor, item_dict, 6, batch_size=16)
    out["neg_neighbor"] = {"seed": 99, "calls": [[it, s.neg_neighbor(it)] for it in [0, 5, 17, 59, 30]]}
    random.seed(123)
    np.random.seed(123)
    s = ref_sampler.Sampler({}, sess_d, time_d, impressions, item_dict, 6, batch_size=16)
    out["neg_impression"] = {"seed": 123,
                             "calls": [[sid, s.neg_neighbor_from_impre(sid)] for sid in sorted(impressions)[:12]]}

    # (5) bucketized over 0..2000 s and a few large values
    secs = list(range(0, 2001)) + [5000, 86399, 10 ** 6]
    out["bucketized"] = {"seconds": secs, "bucket": [int(ref_sampler.bucketized(x)) for x in secs]}

    with open(os.path.join(HERE, "reference_sampler.json"), "w") as fh:
        json.dump(_jsonable(out), fh)

    # (6) cau_metrics
    r = np.random.RandomState(0)
    cases = []
    preds = r.standard_normal((12, 50)).astype(np.float32)
    labels = r.randint(0, 50, size=12)
    cases.append({"preds": preds, "labels": labels, "cutoff": 20})
    cases.append({"preds": np.array([[.1, .9, .3, .3], [.5, .4, .3, .2]], dtype=np.float32),
                  "labels": np.array([2, 0]), "cutoff": 2})
    tie = np.zeros((3, 30), dtype=np.float32)               # all-equal rows: rank 1 for every label
    tie[1, :25] = 1.0                                       # label in the low group: rank 26 > cutoff
    cases.append({"preds": tie, "labels": np.array([7, 29, 0]), "cutoff": 20})
    big = r.standard_normal((4, 500)).astype(np.float32)
    cases.append({"preds": big, "labels": np.array([0, 499, 250, 3]), "cutoff": 20})
    mout = []
    for c in cases:
        rec, mrr, ndcg = ref_util.cau_metrics(c["preds"], c["labels"], c["cutoff"])
        mout.append({"preds": c["preds"], "labels": c["labels"], "cutoff": c["cutoff"],
                     "recall": [bool(x) for x in rec], "mrr": [float(x) for x in mrr],
                     "ndcg": [float(x) for x in ndcg]})
    with open(os.path.join(HERE, "reference_metrics.json"), "w") as fh:
        json.dump(_jsonable(mout), fh)
    neighbor_fixture()
    print("wrote reference_sampler.json, reference_metrics.json, reference_neighbor.json")


def neighbor_fixture():
    """(7) generate_neighbor.get_neighbor (generate_neighbor.py:7-21): the +-100 publish-time window incl. the item itself,
    and the pad branch (reachable for 51 <= N < 100: random.sample from the global `random` state)."""
    sys.path.insert(0, os.path.join(REF, "data_process"))
    import generate_neighbor as ref_nb   # noqa
    sys.path.remove(os.path.join(REF, "data_process"))
    r = np.random.RandomState(42)
    cases = []
    pt = r.randint(0, 4000, size=300)                                   # ints, many ties
    cases.append({"name": "ints300", "kind": "int", "publish_time": pt.tolist(), "seed": 1})
    base = datetime.datetime(2017, 1, 1)
    dts = [base + datetime.timedelta(minutes=int(m)) for m in r.randint(0, 600, size=130)]
    cases.append({"name": "datetimes130", "kind": "datetime", "publish_time": [d.isoformat() for d in dts], "seed": 2})
    pt60 = r.randint(0, 10 ** 6, size=60)
    cases.append({"name": "ints60_pad", "kind": "int", "publish_time": pt60.tolist(), "seed": 77})
    out = []
    for c in cases:
        arg = [datetime.datetime.fromisoformat(x) for x in c["publish_time"]] if c["kind"] == "datetime" else list(c["publish_time"])
        random.seed(c["seed"])
        d = ref_nb.get_neighbor(arg)
        out.append(dict(c, keys=[int(k) for k in d.keys()], lists=[[int(x) for x in v] for v in d.values()]))
    with open(os.path.join(HERE, "reference_neighbor.json"), "w") as fh:
        json.dump(out, fh)


if __name__ == "__main__":
    main()
