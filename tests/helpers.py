"""Shared helpers for tests: fixture loading (JSON -> the reference's dict/datetime form)."""
import datetime
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_sampler_fixture():
    with open(os.path.join(GOLD, "reference_sampler.json")) as fh:
        fx = json.load(fh)

    def fix_split(d):
        len_dict = {int(k): list(v) for k, v in d["len_dict"].items()}
        times = {}
        for k, tl in d["session_time_dict"].items():
            times[k] = [dict(t, click_t=datetime.datetime.fromisoformat(t["click_t"]),
                             publish_t=datetime.datetime.fromisoformat(t["publish_t"])) for t in tl]
        return len_dict, d["session_dict"], times

    ds = fx["dataset"]
    fx["train"] = fix_split(ds["train"])
    fx["test"] = fix_split(ds["test"])
    fx["item_dict"] = {int(k): int(v) for k, v in ds["item_dict"].items()}
    fx["neighbor"] = {int(k): v for k, v in ds["neighbor"].items()}
    fx["impressions"] = {int(k): v for k, v in ds["impressions"].items()}
    fx["neg_num"] = ds["neg_num"]
    return fx


def write_reference_fold(dirname, fold, foldnum=1, with_active=True):
    """Write a SynthFold to disk in the pickle layout `data_partition` reads (util.py:20-56; files written by
    globo_preprocess.py:295-364): len_dict / session_dict / session_time_dict for train and test, item_dict,
    neighbor, content_weight, publish_time (= [datetimes, MWDHM]), the category pickle and item_freq_dict_norm."""
    import pickle
    f = str(foldnum)

    def dump(name, obj):
        with open(os.path.join(dirname, name), "wb") as fh:
            pickle.dump(obj, fh)

    for split, store in (("train", fold.train), ("test", fold.test)):
        len_d, sess, times = fold.to_dicts(store, with_active=with_active)
        dump("len_dict_%s%s.pkl" % (split, f), len_d)
        dump("session_dict_%s_%s.pkl" % (split, f), sess)
        dump("session_time_dict_%s%s.pkl" % (split, f), times)
    item_dict = {10_000 + i: i + 1 for i in range(fold.n_items)}              # original article id -> 1-based count
    dump("item_dict_%s.txt" % f, item_dict)
    dump("neighbor_%s.txt" % f, fold.neighbor_dict())
    dump("content_weight_%s.txt" % f, fold.content)
    pub = [t.astype("datetime64[s]").item() for t in fold.publish_ts]
    dump("publish_time_%s.txt" % f, [pub, fold.mwdhm])
    dump("item_freq_dict_norm_%s.txt" % f, {10_000 + i: 1.0 / fold.n_items for i in range(fold.n_items)})
    dump("articles_category.pkl", {10_000 + i: int(c) for i, c in enumerate(fold.category)})
    return item_dict
