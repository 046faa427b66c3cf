"""Synthetic folds in the TCAR dataset contract (there is no network for Globo / Adressa / MIND).

Shapes follow SURVEY.md §8(d): Zipf item popularity, session length 2 + Geometric, prefix augmentation exactly
as ``split_seq`` (globo_preprocess.py:195-213), click times over 16 days, log-uniform gaps.  The generator
writes straight into a `SessionStore` (flat arrays); `to_dicts` materialises the reference's dict/pickle
form for small folds (tests, CLI demos).
"""
from __future__ import annotations

import datetime as _dt
from typing import Dict, Optional, Tuple

import numpy as np

from .data import SessionStore, bucketize_seconds

_EPOCH0 = np.datetime64("2017-10-01T00:00:00")  # the Globo log starts 2017-10-01


def _fields(ts: np.ndarray) -> np.ndarray:
    """datetime64[s] -> int array [...,5] = month(1-12), day(1-31), isoweekday(1-7), hour(0-23), minute(0-59)."""
    ts = ts.astype("datetime64[s]")
    days = ts.astype("datetime64[D]")
    months = ts.astype("datetime64[M]")
    month = (months.astype(np.int64) % 12) + 1
    day = (days - months.astype("datetime64[D]")).astype(np.int64) + 1
    wd = (days.astype(np.int64) + 3) % 7 + 1            # 1970-01-01 was a Thursday (isoweekday 4)
    secs = (ts - days.astype("datetime64[s]")).astype(np.int64)
    return np.stack([month, day, wd, secs // 3600, (secs // 60) % 60], -1)


class SynthFold:
    """One synthetic fold: catalog + train/test stores (+ optional negative sources)."""

    def __init__(self, n_items=46033, dim=250, n_train=100000, n_test=10000, seed=2020, p_len=0.45,
                 zipf_s=1.1, max_clicks=41, active_t=False, same_click_time=False, n_categories=300,
                 content_scale=0.5, lean=False):
        """lean=True (multi-million-item catalogs): the content table is drawn in fp32 row chunks (no fp64 copy of the
        whole table) and the orig-id dictionary is not materialised (`item_dict` is None)."""
        rng = np.random.RandomState(seed)
        self.n_items, self.dim = n_items, dim
        # catalog ------------------------------------------------------------------------------------------
        if lean:
            g32 = np.random.default_rng(seed)
            content = np.empty((n_items + 1, dim), dtype=np.float32)
            for lo in range(0, n_items + 1, 1 << 20):
                hi = min(n_items + 1, lo + (1 << 20))
                content[lo:hi] = g32.standard_normal((hi - lo, dim), dtype=np.float32) * np.float32(content_scale)
        else:
            content = (rng.standard_normal((n_items + 1, dim)) * content_scale).astype(np.float32)
        content[0] = 0.0                                   # globo_preprocess.py:315
        self.content = content
        pub_ts = _EPOCH0 + rng.randint(-30 * 86400, 16 * 86400, size=n_items).astype("timedelta64[s]")
        f = _fields(pub_ts)
        self.publish_ts = pub_ts
        self.mwdhm = np.stack([f[:, 0], f[:, 1], f[:, 2], f[:, 3] + 1, f[:, 4] + 1], -1).astype(np.int32)
        self.category = rng.randint(0, n_categories, size=n_items).astype(np.int32)   # category of 0-based item
        w = 1.0 / np.arange(1, n_items + 1, dtype=np.float64) ** zipf_s
        self._cdf = np.cumsum(w / w.sum())
        self._perm = rng.permutation(n_items).astype(np.int64)   # popularity rank -> item id (0-based)
        self.item_dict = None if lean else {int(10_000 + i): int(i + 1) for i in range(n_items)}   # orig id -> 1-based id
        self.train = self._sessions(rng, n_train, p_len, max_clicks, active_t, same_click_time, 0)
        self.test = self._sessions(rng, n_test, p_len, max_clicks, active_t, same_click_time, 50_000_000)

    # -----------------------------------------------------------------------------------------------------
    def _sessions(self, rng, n_examples, p_len, max_clicks, active_t, same_click_time, tag) -> SessionStore:
        """Draw raw sessions until the prefix-augmented example count reaches n_examples."""
        n_raw = max(1, int(n_examples / max(1.0, (1.0 / p_len))) + 8)
        L = np.minimum(2 + rng.geometric(p_len, size=n_raw) - 1, max_clicks)
        while int((L - 1).sum()) < n_examples:
            extra = np.minimum(2 + rng.geometric(p_len, size=n_raw) - 1, max_clicks)
            L = np.concatenate([L, extra])
        # keep just enough raw sessions
        cum = np.cumsum(L - 1)
        n_keep = int(np.searchsorted(cum, n_examples) + 1)
        L = L[:n_keep]
        tot = int(L.sum())
        raw_off = np.zeros(n_keep + 1, dtype=np.int64)
        np.cumsum(L, out=raw_off[1:])
        ranks = np.searchsorted(self._cdf, rng.random_sample(tot)).clip(0, self.n_items - 1)
        items0 = self._perm[ranks]                                              # 0-based
        start = rng.randint(0, 16 * 86400 - 3600 * 12, size=n_keep)
        gaps = np.exp(rng.uniform(0.0, np.log(1800.0), size=tot)).astype(np.int64)
        if same_click_time:                                                     # MIND: one click time per session
            gaps[:] = 0
        first = np.zeros(tot, dtype=bool)
        first[raw_off[:-1]] = True
        gaps[first] = 0
        csum = np.cumsum(gaps)
        rel = csum - np.repeat(csum[raw_off[:-1]], L)
        click_s = np.repeat(start, L) + rel
        click_ts = _EPOCH0 + click_s.astype("timedelta64[s]")
        cf = _fields(click_ts)
        clk_all = np.stack([cf[:, 0] - 1, cf[:, 1] - 1, cf[:, 2] - 1, cf[:, 3], cf[:, 4]], -1)
        pub_all = self.mwdhm[items0]
        if active_t:
            act = np.clip(np.exp(rng.normal(3.5, 1.2, size=tot)), 1, 1023)
            if same_click_time:
                act[:] = 1.0                                                    # mind_preprocess.py:22
        else:
            act = np.zeros(tot)
        # prefix augmentation: session of L clicks -> examples with input length 1..L-1 (split_seq)
        ex_raw = np.repeat(np.arange(n_keep), L - 1)
        ex_len = np.concatenate([np.arange(l, 1, -1) for l in L]) if n_keep else np.zeros(0, np.int64)
        ex_raw, ex_len = ex_raw[:n_examples], ex_len[:n_examples]              # ex_len = clicks incl. label
        off = np.zeros(len(ex_len) + 1, dtype=np.int64)
        np.cumsum(ex_len, out=off[1:])
        src = np.repeat(raw_off[ex_raw], ex_len) + (np.arange(off[-1]) - np.repeat(off[:-1], ex_len))
        nxt = np.minimum(src + 1, tot - 1)
        delta = np.where(np.arange(off[-1]) - np.repeat(off[:-1], ex_len) < np.repeat(ex_len, ex_len) - 1,
                         (click_s[nxt] - click_s[src]) % 86400, 0)
        keys = ["%d_%d" % (tag + r, l - 1) for r, l in zip(ex_raw.tolist(), ex_len.tolist())] \
            if len(ex_len) <= 200000 else None
        store = SessionStore(off, items0[src] + 1, pub_all[src], clk_all[src],
                             bucketize_seconds(act[src]) if active_t else np.zeros(len(src), np.uint8),
                             bucketize_seconds(delta), keys, ex_raw.astype(np.int64) + tag)
        store.click_ts = click_ts[src]
        store.active_s = act[src]
        return store

    # ------------------------------------------------------------------------------- negative-source builders
    def neighbor_dict(self, k=100) -> Dict[int, np.ndarray]:
        """`get_neighbor` (generate_neighbor.py:7-21) over the fold's publish times: host/data.py:build_neighbor — `k` earlier
        items, the item itself and `k - 1` later ones in publish-time order."""
        from .data import build_neighbor
        return build_neighbor(self.publish_ts, window=k)

    def impression_dict(self, store: SessionStore, lo=20, hi=60, unknown=0.1, seed=7) -> Dict[int, list]:
        """MIND-style impression lists (mind_preprocess.py:62-69,85): per raw session id 20-60 ORIGINAL article ids, about
        10 % of them outside the catalog (never-clicked candidates: sampler.py:124 skips those)."""
        rng = np.random.RandomState(seed)
        out = {}
        for sid in np.unique(store.impression_key).tolist():
            n = int(rng.randint(lo, hi + 1))
            ids = 10_000 + rng.randint(0, self.n_items, n)
            ids[rng.random_sample(n) < unknown] += 50_000_000
            out[int(sid)] = ids.tolist()
        return out

    def to_dicts(self, store: SessionStore, with_active=False, examples=None) -> Tuple[dict, dict, dict]:
        """Materialise (len_dict, session_dict, session_time_dict) in the reference's pickle form (`examples`: only these rows
        of the store)."""
        len_dict, sess, times = {}, {}, {}
        for e in (range(store.n) if examples is None else [int(x) for x in examples]):
            o, o2 = int(store.off[e]), int(store.off[e + 1])
            key = store.keys[e]
            ids = store.items[o:o2].tolist()
            len_dict.setdefault(len(ids) - 1, []).append(key)
            sess[key] = ids
            tl = []
            for j in range(o, o2):
                d = {"click_t": store.click_ts[j].astype("datetime64[s]").item(),
                     "publish_t": self.publish_ts[store.items[j] - 1].astype("datetime64[s]").item()}
                if with_active:
                    d["active_t"] = float(store.active_s[j])
                tl.append(d)
            times[key] = tl
        return len_dict, sess, times

    def model_args(self, **over) -> dict:
        """The `args` dict main.py:24-43 hands to Seq2SeqAttNN (flags + catalog entries)."""
        a = dict(batch_size=512, lr=0.001, epoch=1, maxlen=20, neg_num=20, hidden_size=self.dim,
                 time_hidden_size=64, max_grad=150, stddev=0.05, emb_stddev=0.002, dropout_rate=0.5,
                 l2_emb=0.0, is_print=False, foldnum=0)
        a.update(itemnum=self.n_items, reverse_item={i: 10_000 + i for i in range(self.n_items)},
                 category_id={10_000 + i: int(c) for i, c in enumerate(self.category)},
                 item_freq_dict_norm=None, publish_time=[None] * self.n_items,
                 publish_time_MWDHM=self.mwdhm, content_emb=self.content)
        a.update(over)
        return a
