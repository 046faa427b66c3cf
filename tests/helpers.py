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
