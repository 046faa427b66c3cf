"""Host-side batch producer with the reference's ``Sampler`` interface (sampler.py:23-116).

Same constructor arguments, ``has_next()`` / ``next_batch()`` and the same use of the global ``random`` /
``numpy.random`` streams (so a seeded run forms the same batches as the reference), but the per-click Python
loop of ``sampler.py:67-111`` is replaced by vectorised gathers over a `SessionStore` that is built once per
dataset and cached.  ``next_batch_arrays()`` is what the HIP training loop consumes (int32 arrays, ready for
one H2D copy); ``next_batch()`` re-nests them into the reference's 6-tuple of lists.

The two call-site toggles the reference leaves as comments are explicit modes here:
  gap_mode: "active_t" (sampler.py:87, shipped) | "click_delta" (sampler.py:91-94, Globo)
  neg_mode: "uniform" (sampler.py:98-99, shipped) | "neighbor" (:97,:133-140) | "impression" (:96,:118-131)

``neg_fast=True`` draws the neighbour / impression negatives with vectorised numpy from a CSR of the source lists (same
distribution and rules — neighbour picks differ from the label; impression mode makes at most 21 tries, keeps the
candidates present in ``item_dict`` and pads with uniform draws — but NOT the reference's ``random.choice`` call
order, which costs ~5 ms of Python per batch of 512 x 20 and would starve a 0.7-ms device step).
"""
from __future__ import annotations

import random
from typing import Dict, Optional

import numpy as np

from .data import SessionStore

_STORE_CACHE: Dict[int, SessionStore] = {}
_CSR_CACHE: Dict[tuple, tuple] = {}          # (id(neighbor_dict), mode) -> CSR of the negative source (built once per dataset)


def store_for(session_dict, session_time_dict) -> SessionStore:
    """Tensorise once per dataset object (the trainer builds a new Sampler every epoch, model_combine.py:204)."""
    key = id(session_dict)
    st = _STORE_CACHE.get(key)
    if st is None or st.n != len(session_dict):
        st = SessionStore.from_dicts(session_dict, session_time_dict)
        _STORE_CACHE[key] = st
    return st


class Sampler(object):
    def __init__(self, len_dict, session_dict, session_time_dict=None, neighbor_dict=None, item_dict=None,
                 neg_num=None, batch_size=1024, gap_mode="active_t", neg_mode="uniform", store=None,
                 verbose=True, neg_fast=False):
        if verbose:
            print('Sampler init begin...')
        self.session_num = len(session_dict) if session_dict is not None else (store.n if store else 0)
        self.batch_size = batch_size
        self.batch_num = 0
        self.batch_i = 0
        self.neighbor_dict = neighbor_dict
        self.item_dict = item_dict
        if item_dict is not None:
            self.item_num = len(item_dict)
        self.neg_num = neg_num
        self.len_dict = len_dict
        self.session_dict = session_dict
        self.session_time_dict = session_time_dict
        self.gap_mode, self.neg_mode = gap_mode, neg_mode
        self.neg_fast = bool(neg_fast)
        self._csr = None
        if neg_mode not in ("uniform", "neighbor", "impression"):
            raise ValueError("neg_mode must be uniform | neighbor | impression")
        if gap_mode not in ("active_t", "click_delta"):
            raise ValueError("gap_mode must be active_t | click_delta")
        self.store = store if store is not None else store_for(session_dict, session_time_dict)
        self.has_time = bool(session_time_dict) or store is not None
        self.session_id_batches = []
        for _slen, ids in len_dict.items():
            random.shuffle(ids)                      # in place, like the reference (persists across epochs)
            n = len(ids)
            start = 0
            while n - start > batch_size:            # strict: a bucket of exactly batch_size is one batch
                self.session_id_batches.append(ids[start:start + batch_size])
                start += batch_size
            if n - start:
                self.session_id_batches.append(ids[start:])
        random.shuffle(self.session_id_batches)
        self.batch_num = len(self.session_id_batches)
        if verbose:
            print('Sampler init finished, batch size : {}, # batch: {}.'.format(self.batch_size, self.batch_num))

    def has_next(self):
        return self.batch_i < self.batch_num

    # ------------------------------------------------------------------ negatives
    def _negatives(self, keys, labels0, idx=None) -> Optional[np.ndarray]:
        if not self.neighbor_dict or not self.has_time:        # sampler.py:72,95: only inside the time branch
            return None
        K, B = self.neg_num, len(keys)
        if self.neg_mode == "uniform":
            # K scalar draws per session in the reference == one vector draw from the same legacy stream
            return np.random.randint(0, self.item_num, size=(B, K)).astype(np.int32)
        # impression mode: the session id of an example — parsed from its "sid_len" key (sampler.py:96) or, for stores
        # with integer example ids, taken from the store's impression_key column
        ik = self.store.impression_key if (idx is not None and self.store is not None) else None
        if self.neg_fast:
            sids = ik[idx] if (ik is not None and self.neg_mode == "impression") else None
            return self._negatives_fast(keys, np.asarray(labels0, dtype=np.int64), sids)
        out = np.empty((B, K), dtype=np.int32)
        for b, key in enumerate(keys):
            if self.neg_mode == "neighbor":
                out[b] = self.neg_neighbor(int(labels0[b]))
            else:
                sid = int(ik[idx[b]]) if ik is not None else int(str(key).split('_')[0])
                out[b] = self.neg_neighbor_from_impre(sid)
        return out

    # vectorised variants ---------------------------------------------------------------------------------------
    def _source_csr(self):
        """neighbor_dict as CSR over its keys: (key -> slot, offsets, flat candidate ids).  Impression candidates are
        mapped through item_dict once (original article id -> 0-based item id, -1 when absent)."""
        ck = (id(self.neighbor_dict), self.neg_mode, len(self.neighbor_dict))
        if self._csr is None:
            self._csr = _CSR_CACHE.get(ck)
        if self._csr is None:
            keys = list(self.neighbor_dict.keys())
            slot = {k: i for i, k in enumerate(keys)}
            lens = np.fromiter((len(self.neighbor_dict[k]) for k in keys), dtype=np.int64, count=len(keys))
            off = np.zeros(len(keys) + 1, dtype=np.int64)
            np.cumsum(lens, out=off[1:])
            flat = np.empty(int(off[-1]), dtype=np.int64)
            for i, k in enumerate(keys):
                c = self.neighbor_dict[k]
                if self.neg_mode == "impression":
                    flat[off[i]:off[i + 1]] = [self.item_dict.get(x, 0) - 1 for x in c]
                else:
                    flat[off[i]:off[i + 1]] = c
            arr = None
            if keys and all(isinstance(k, (int, np.integer)) and 0 <= k < (1 << 26) for k in keys):
                arr = np.full(int(max(keys)) + 1, -1, dtype=np.int64)          # dense key -> slot map (vectorised lookup)
                arr[np.asarray(keys, dtype=np.int64)] = np.arange(len(keys))
            self._csr = _CSR_CACHE[ck] = (slot, off, flat, arr, {})
        return self._csr

    def _negatives_fast(self, keys, labels0, sids=None) -> np.ndarray:
        slot, off, flat, arr, kcache = self._source_csr()
        K, B = self.neg_num, len(keys)
        if self.neg_mode == "neighbor":
            if arr is not None and labels0.max() < len(arr):
                sl = arr[labels0]
                if (sl < 0).any():
                    raise KeyError("label without a neighbour list")
            else:
                sl = np.fromiter((slot[int(x)] for x in labels0), dtype=np.int64, count=B)
        elif sids is not None and arr is not None and int(np.max(sids)) < len(arr):
            sl = arr[np.asarray(sids, dtype=np.int64)]
            if (sl < 0).any():
                raise KeyError("session without an impression list")
        else:
            sl = np.empty(B, dtype=np.int64)
            for b, k in enumerate(keys):               # session key "sid_len" -> slot, parsed once per key
                v = kcache.get(k)
                if v is None:
                    v = kcache[k] = slot[int(str(k).split('_')[0])]
                sl[b] = v
        lo, cnt = off[sl], off[sl + 1] - off[sl]
        if (cnt <= 0).any():
            raise KeyError("empty negative source list")
        if self.neg_mode == "neighbor":                   # K picks from the label's list, each != label (sampler.py:133-140)
            pick = flat[lo[:, None] + (np.random.random_sample((B, K)) * cnt[:, None]).astype(np.int64)]
            bad = pick == labels0[:, None]
            while bad.any():
                r, c = np.nonzero(bad)
                pick[r, c] = flat[lo[r] + (np.random.random_sample(len(r)) * cnt[r]).astype(np.int64)]
                bad = pick == labels0[:, None]
            return pick.astype(np.int32)
        # impression (sampler.py:118-131): at most 21 tries, keep candidates that are catalog items, pad with uniform draws
        tries = flat[lo[:, None] + (np.random.random_sample((B, 21)) * cnt[:, None]).astype(np.int64)]
        valid = tries >= 0
        order = np.argsort(~valid, axis=1, kind="stable")                 # valid tries first, in try order
        tries = np.take_along_axis(tries, order, 1)[:, :K]
        nvalid = np.minimum(valid.sum(1), K)
        fill = np.arange(K)[None, :] >= nvalid[:, None]
        tries[fill] = np.random.randint(0, self.item_num, size=int(fill.sum()))
        return tries.astype(np.int32)

    def neg_neighbor_from_impre(self, sessionid):
        cand = self.neighbor_dict[sessionid]
        neg, tries = [], 0
        while len(neg) < self.neg_num:
            tries += 1
            pick = random.choice(cand)
            if pick in self.item_dict:
                neg.append(self.item_dict[pick] - 1)
            if tries > 20:
                break
        if len(neg) < self.neg_num:
            neg.extend(np.random.randint(0, self.item_num) for _ in range(self.neg_num - len(neg)))
        return neg

    def neg_neighbor(self, itemid):
        cand = self.neighbor_dict[itemid]
        neg = []
        while len(neg) < self.neg_num:
            pick = random.choice(cand)
            if pick != itemid:
                neg.append(pick)
        return neg

    # -------------------------------------------------------------------- batches
    def batch_indices(self, i: int) -> np.ndarray:
        """store row indices of the examples of batch i (the device sampler forms the batch from these alone)"""
        keys = self.session_id_batches[i]
        st = self.store
        if st.key_index is not None and not isinstance(keys[0], (int, np.integer)):
            return np.fromiter((st.key_index[k] for k in keys), dtype=np.int64, count=len(keys))
        if st.key_index is not None and keys[0] in st.key_index:
            return np.fromiter((st.key_index[k] for k in keys), dtype=np.int64, count=len(keys))
        return np.asarray(keys, dtype=np.int64)              # store-native integer example ids

    def next_batch_arrays(self) -> Dict[str, np.ndarray]:
        keys = self.session_id_batches[self.batch_i]
        st = self.store
        idx = self.batch_indices(self.batch_i)
        arr = st.batch_arrays(idx, self.gap_mode)
        arr["neg"] = self._negatives(keys, arr["label"], idx)
        arr["keys"] = keys
        self.batch_i += 1
        return arr

    def next_batch(self):
        a = self.next_batch_arrays()
        B = len(a["label"])
        if self.has_time:
            pub = tuple(a[k].tolist() for k in ("pm", "pd", "pw", "ph", "pmi"))
            clk = tuple(a[k].tolist() for k in ("cmo", "cd", "cw", "ch", "cmi"))
            gap = a["gap"].tolist()
        else:
            pub, clk, gap = tuple([] for _ in range(5)), tuple([] for _ in range(5)), [[] for _ in range(B)]
        neg = a["neg"].tolist() if a["neg"] is not None else [[] for _ in range(B)]
        return a["seq"].tolist(), a["label"].tolist(), pub, clk, neg, gap
