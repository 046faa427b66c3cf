"""Dataset contract of the TCAR trainer and its one-time tensorisation.

The reference trainer reads a set of pickles (``util.py:20-56``; written by
``data_process/globo_preprocess.py:295-364`` and its Adressa / MIND twins):

  len_dict[T]            -> list of session keys whose INPUT length is T
  session_dict[key]      -> [item ids (1-based) of the T inputs ..., label id]
  session_time_dict[key] -> list of T+1 dicts {'click_t': datetime, 'publish_t': datetime,
                            'active_t': seconds (Adressa/MIND only), ...}
  item_dict[orig_id]     -> 1-based dense id;  content_weight [(N+1), d] (row 0 zeros);
  publish_time = (list of datetime, int array [N,5] = month, day, isoweekday, hour+1, minute+1)

`SessionStore` converts the three session dicts ONCE into flat integer arrays (CSR over clicks) so that
the per-click Python loop of ``sampler.py:67-111`` disappears from the training loop: a batch is then a
handful of vectorised gathers.  All per-click features the sampler derives are precomputed here with the
same formulas (sampler.py:81-87,105-109; gap variants :87 and :91-94).
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, List, Optional, Sequence

import numpy as np

GAP_OOB = 11  # bucketized() returns 11 for >= 1024 s (sampler.py:18-21); the duration table has 11 rows


def bucketize_seconds(sec: np.ndarray) -> np.ndarray:
    """Vectorised ``bucketized`` (sampler.py:18-21): searchsorted([0..10], log2(s+1), side='left')."""
    sec = np.asarray(sec, dtype=np.float64)
    return np.searchsorted(np.arange(0, 11), np.log2(sec + 1)).astype(np.uint8)


def _dt_fields(dt) -> tuple:
    return (dt.month, dt.day, dt.isoweekday(), dt.hour, dt.minute)


class SessionStore:
    """Flat, integer-only view of (len_dict, session_dict, session_time_dict).

    Arrays (n = number of examples, c = total clicks incl. labels):
      off[n+1]      int64   CSR offsets into the per-click arrays
      items[c]      int32   1-based item ids; the last click of an example is its label
      pub[c,5]      uint8   publish month, day, isoweekday, hour+1, minute+1   (sampler.py:81-85)
      clk[c,5]      uint8   click month-1, day-1, isoweekday-1, hour, minute  (sampler.py:105-109)
      gap_active[c] uint8   bucketized(active_t) (sampler.py:87) or 0 when the field is absent
      gap_delta[c]  uint8   bucketized(seconds to the NEXT click of the example) (sampler.py:91-94)
      key_index     dict    session key -> example index
    """

    def __init__(self, off, items, pub, clk, gap_active, gap_delta, keys: Optional[Sequence] = None,
                 impression_key: Optional[np.ndarray] = None):
        self.off = np.asarray(off, dtype=np.int64)
        self.items = np.asarray(items, dtype=np.int32)
        self.pub = np.asarray(pub, dtype=np.uint8)
        self.clk = np.asarray(clk, dtype=np.uint8)
        self.gap_active = np.asarray(gap_active, dtype=np.uint8)
        self.gap_delta = np.asarray(gap_delta, dtype=np.uint8)
        self.keys = list(keys) if keys is not None else None
        self.key_index = {k: i for i, k in enumerate(self.keys)} if self.keys is not None else None
        self.impression_key = impression_key   # int id parsed from "sid_len" keys (sampler.py:96)
        self.n = len(self.off) - 1
        self.in_len = (self.off[1:] - self.off[:-1] - 1).astype(np.int32)

    @classmethod
    def from_dicts(cls, session_dict: Dict, session_time_dict: Optional[Dict]) -> "SessionStore":
        keys = list(session_dict.keys())
        lens = np.fromiter((len(session_dict[k]) for k in keys), dtype=np.int64, count=len(keys))
        off = np.zeros(len(keys) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        c = int(off[-1])
        items = np.empty(c, dtype=np.int32)
        pub = np.zeros((c, 5), dtype=np.uint8)
        clk = np.zeros((c, 5), dtype=np.uint8)
        act = np.zeros(c, dtype=np.float64)
        delta = np.zeros(c, dtype=np.float64)
        has_active = False
        for e, k in enumerate(keys):
            o = int(off[e])
            sess = session_dict[k]
            items[o:o + len(sess)] = sess
            if session_time_dict:
                times = session_time_dict[k]
                prev = None
                for j, t in enumerate(times):
                    p, cdt = t["publish_t"], t["click_t"]
                    pm, pd_, pw, ph, pmi = _dt_fields(p)
                    pub[o + j] = (pm, pd_, pw, ph + 1, pmi + 1)
                    cm, cd, cw, ch, cmi = _dt_fields(cdt)
                    clk[o + j] = (cm - 1, cd - 1, cw - 1, ch, cmi)
                    if "active_t" in t:
                        act[o + j] = t["active_t"]
                        has_active = True
                    if prev is not None:
                        delta[o + j - 1] = (cdt - prev).seconds      # timedelta.seconds, as sampler.py:92
                    prev = cdt
        ga = bucketize_seconds(act) if has_active else np.zeros(c, dtype=np.uint8)
        gd = bucketize_seconds(delta)
        imp = None
        try:
            imp = np.array([int(str(k).split("_")[0]) for k in keys], dtype=np.int64)
        except ValueError:
            imp = None
        return cls(off, items, pub, clk, ga, gd, keys, imp)

    # ------------------------------------------------------------------ batches
    def batch_arrays(self, idx: np.ndarray, gap_mode: str = "active_t") -> Dict[str, np.ndarray]:
        """Feed arrays for the examples `idx` (all of the same input length T)."""
        idx = np.asarray(idx, dtype=np.int64)
        T = int(self.in_len[idx[0]])
        if not np.all(self.in_len[idx] == T):
            raise ValueError("a batch must hold sessions of one input length (sampler.py:40-49)")
        base = self.off[idx]
        pos = base[:, None] + np.arange(T, dtype=np.int64)[None, :]
        last = base + T - 1                      # last INPUT click (sampler.py:86,105-109)
        pub = self.pub[pos]                      # [B,T,5]
        gap = (self.gap_active if gap_mode == "active_t" else self.gap_delta)[pos]
        return {
            "seq": self.items[pos].astype(np.int32),
            "label": (self.items[base + T] - 1).astype(np.int32),
            "pm": pub[..., 0].astype(np.int32), "pd": pub[..., 1].astype(np.int32),
            "pw": pub[..., 2].astype(np.int32), "ph": pub[..., 3].astype(np.int32),
            "pmi": pub[..., 4].astype(np.int32),
            "cmo": self.clk[last, 0].astype(np.int32), "cd": self.clk[last, 1].astype(np.int32),
            "cw": self.clk[last, 2].astype(np.int32), "ch": self.clk[last, 3].astype(np.int32),
            "cmi": self.clk[last, 4].astype(np.int32),
            "gap": gap.astype(np.int32),
        }


def build_neighbor(publish_time, window: int = 100) -> Dict[int, np.ndarray]:
    """`get_neighbor` (generate_neighbor.py:7-21): the negative source of `Sampler.neg_neighbor` (sampler.py:133-140).
    Items are ordered by publish time (`np.argsort` of the array as given, numpy's default sort — ties fall where the
    reference's call puts them); the item at position p of that order gets positions [p - window, p + window): `window`
    earlier items, ITSELF, and `window - 1` later ones, as a numpy slice of the order (clipped at both ends).  Lists shorter
    than `window` (only when the catalog has fewer than `window` items) are padded with `random.sample(range(N - 1), ...)`
    from the global `random` state, in dictionary order (generate_neighbor.py:18-20).  Keys and values are 0-based item
    positions, as `neg_neighbor(label0)` expects (sampler.py:97)."""
    import random
    order = np.argsort(np.array(publish_time))
    n = len(publish_time)
    lo = np.maximum(np.arange(n) - window, 0)
    hi = np.minimum(np.arange(n) + window, n)
    out = {item: order[a:b] for item, a, b in zip(order, lo.tolist(), hi.tolist())}
    for item, lst in out.items():
        short = window - len(lst)
        if short > 0:
            out[item] = np.append(lst, np.array(random.sample(range(0, n - 1), short)))
    return out


def load_fold(fname: str, foldnum, neighbor_path: Optional[str] = None):
    """Counterpart of ``data_partition`` (util.py:20-56) without the hard-coded author path (util.py:47):
    the negative-source pickle is `neighbor_path` if given, else ``<fname>neighbor_<fold>.txt`` when it
    exists (util.py:48), else None.  The unused train/test_session tuples (util.py:30-44) are not loaded."""
    f = str(foldnum)

    def ld(name):
        with open(fname + name, "rb") as fh:
            return pickle.load(fh)

    train = (ld("len_dict_train" + f + ".pkl"), ld("session_dict_train_" + f + ".pkl"),
             ld("session_time_dict_train" + f + ".pkl"))
    test = (ld("len_dict_test" + f + ".pkl"), ld("session_dict_test_" + f + ".pkl"),
            ld("session_time_dict_test" + f + ".pkl"))
    item_dict = ld("item_dict_" + f + ".txt")
    neighbor = None
    if neighbor_path:
        with open(neighbor_path, "rb") as fh:
            neighbor = pickle.load(fh)
    elif os.path.exists(fname + "neighbor_" + f + ".txt"):
        neighbor = ld("neighbor_" + f + ".txt")
    content = ld("content_weight_" + f + ".txt")
    publish_time = ld("publish_time_" + f + ".txt")
    return train, test, item_dict, neighbor, content, publish_time, None
