"""HBM-resident session store + negative sources, and batch formation on the device (csrc/sampler.hip, tcar_form_batch).

The host keeps what `Sampler.__init__` does (sampler.py:40-49: shuffle every length bucket, cut it into batches, shuffle the
batches — the reference's use of the `random` stream, so batch composition is unchanged); the per-click loop of
`next_batch` (sampler.py:67-111) and the negative draws (sampler.py:95-99,118-140) run in one kernel launch per batch from
arrays that were uploaded once:

  store      CSR over clicks + per-click uint8 features (host/data.py SessionStore)
  negatives  "uniform":    nothing to upload
             "neighbor":   CSR over the 0-based item id of `neighbor_dict` (generate_neighbor.py:7-21: the +-100 items
                           adjacent in publish-time order)
             "impression": CSR over the session ids of `neighbor_dict` (mind_preprocess.py:62-69,85), candidates mapped
                           through item_dict to 0-based ids (-1 = not a catalog item, sampler.py:124), plus example -> list

Only the example indices of a batch (4 B per session) cross PCIe.  The negatives follow the reference's rules with a
counter-based generator (rule equivalence is tested in tests/test_gpu_sampler.py; draw-for-draw replay of Python's
generators is what the host sampler is for).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import Batch, NegSrc, Store, check
from .host.data import SessionStore

NEG_MODES = {"uniform": 0, "neighbor": 1, "impression": 2}


def neighbor_csr(neighbor_dict: Dict[int, list], n_items: int):
    """CSR over 0-based item ids; an item without a list gets an empty one (the kernel then falls back to uniform draws)."""
    lens = np.zeros(n_items, dtype=np.int64)
    for k, v in neighbor_dict.items():
        if 0 <= int(k) < n_items:
            lens[int(k)] = len(v)
    off = np.zeros(n_items + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    flat = np.empty(int(off[-1]), dtype=np.int32)
    for k, v in neighbor_dict.items():
        k = int(k)
        if 0 <= k < n_items:
            flat[off[k]:off[k + 1]] = v
    return off, flat


def impression_csr(neighbor_dict: Dict[int, list], item_dict: Dict[int, int], store: SessionStore):
    """CSR over the session ids that own an impression list; candidates as 0-based item ids, -1 when the article is not in
    item_dict (sampler.py:124); slot_of_example maps an example to its session's list."""
    if store.impression_key is None:
        raise ValueError("impression negatives need the session id of every example (SessionStore.impression_key)")
    keys = list(neighbor_dict.keys())
    slot = {int(k): i for i, k in enumerate(keys)}
    lens = np.fromiter((len(neighbor_dict[k]) for k in keys), dtype=np.int64, count=len(keys))
    off = np.zeros(len(keys) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    flat = np.empty(int(off[-1]), dtype=np.int32)
    for i, k in enumerate(keys):
        flat[off[i]:off[i + 1]] = [item_dict.get(x, 0) - 1 for x in neighbor_dict[k]]
    try:
        soe = np.fromiter((slot[int(s)] for s in store.impression_key), dtype=np.int32, count=store.n)
    except KeyError as e:
        raise KeyError("session %s has no impression list" % e)
    return off, flat, soe


class DeviceSampler:
    def __init__(self, engine, store: SessionStore, neg_mode: str = "uniform", neighbor_dict: Optional[dict] = None,
                 item_dict: Optional[dict] = None, seed: int = 2020):
        if neg_mode not in NEG_MODES:
            raise ValueError("neg_mode must be uniform | neighbor | impression")
        self.eng, self.store, self.neg_mode = engine, store, neg_mode
        self.lib, self.dev = engine.lib, engine.dev
        self.seed, self.counter = int(seed) & ((1 << 64) - 1), 0
        up = lambda a, dt: torch.tensor(np.ascontiguousarray(a, dtype=dt), device=self.dev)
        self.t_off, self.t_items = up(store.off, np.int64), up(store.items, np.int32)
        self.t_pub, self.t_clk = up(store.pub, np.uint8), up(store.clk, np.uint8)
        self.t_ga, self.t_gd = up(store.gap_active, np.uint8), up(store.gap_delta, np.uint8)
        self.in_len = store.in_len
        st = Store()
        st.off, st.items, st.pub, st.clk = (t.data_ptr() for t in (self.t_off, self.t_items, self.t_pub, self.t_clk))
        st.gap_active, st.gap_delta, st.n_examples = self.t_ga.data_ptr(), self.t_gd.data_ptr(), store.n
        self.c_store = st
        src = NegSrc()
        src.mode = NEG_MODES[neg_mode]
        if neg_mode == "neighbor":
            off, flat = neighbor_csr(neighbor_dict, engine.geo.N)
            self.t_noff, self.t_nflat = up(off, np.int64), up(flat, np.int32)
            src.off, src.flat, src.n_lists = self.t_noff.data_ptr(), self.t_nflat.data_ptr(), engine.geo.N
        elif neg_mode == "impression":
            off, flat, soe = impression_csr(neighbor_dict, item_dict, store)
            self.t_noff, self.t_nflat, self.t_soe = up(off, np.int64), up(flat, np.int32), up(soe, np.int32)
            src.off, src.flat, src.slot_of_example = self.t_noff.data_ptr(), self.t_nflat.data_ptr(), self.t_soe.data_ptr()
            src.n_lists = len(off) - 1
        self.c_src = src
        self.feed = None
        self.pin = None

    def bytes_resident(self) -> int:
        return sum(t.numel() * t.element_size() for t in vars(self).values() if torch.is_tensor(t))

    def _describe(self, feed: torch.Tensor, B: int, T: int, K: int) -> Batch:
        """C batch descriptor over a formed feed buffer (layout of tcar_form_batch = TcarEngine.upload)"""
        base = feed.data_ptr()
        bt = Batch()
        bt.B, bt.T, bt.K = B, T, K
        n = B * T
        bt.seq = base
        for k in range(5):
            bt.pub[k] = base + 4 * (k + 1) * n
        bt.gap = base + 4 * 6 * n
        bt.cw = base + 4 * 7 * n
        bt.ch = base + 4 * (7 * n + B)
        bt.label = base + 4 * (7 * n + 2 * B)
        bt.neg = (base + 4 * (7 * n + 3 * B)) if K else None
        bt._seq_t = feed[:n]
        bt._keep = feed
        return bt

    def _launch(self, idx_ptr: int, B: int, T: int, K: int, gap_mode: str, counter: int, feed: torch.Tensor, stream) -> None:
        if T > 40:
            raise IndexError("session longer than the 40-row position table (model_combine.py:57)")
        check(self.lib.tcar_form_batch(C.byref(self.eng.dims), C.byref(self.c_store), C.byref(self.c_src), C.c_void_p(idx_ptr),
                                       B, T, K, 1 if gap_mode == "click_delta" else 0, C.c_uint64(self.seed),
                                       C.c_uint64(int(counter)), C.c_void_p(feed.data_ptr()), C.c_void_p(stream.cuda_stream)),
              "tcar_form_batch")

    def form(self, idx: np.ndarray, K: int, gap_mode: str = "active_t", counter: Optional[int] = None) -> Batch:
        """Feed of the batch whose examples are `idx` (all of one input length), formed on the device; returns the C batch
        descriptor for TcarEngine.train_step / eval_step (bt=...).  K = 0: no negatives (evaluation)."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        B = int(idx.shape[0])
        T = int(self.in_len[idx[0]])
        need = 7 * B * T + 3 * B + B * K
        if self.feed is None or self.feed.numel() < need or self.idx_dev.numel() < B:
            n = max(need, 1 << 16)
            self.feed = torch.empty(n, dtype=torch.int32, device=self.dev)
            nb = max(B, 4096)
            self.idx_dev = torch.empty(nb, dtype=torch.int32, device=self.dev)
            self.pin = [torch.empty(nb, dtype=torch.int32).pin_memory() for _ in range(2)]
            self.pin_evt = [torch.cuda.Event(), torch.cuda.Event()]
            self.pin_used, self.pin_i = [False, False], 0
        i = self.pin_i = self.pin_i ^ 1
        if self.pin_used[i]:
            self.pin_evt[i].synchronize()
        self.pin[i][:B].copy_(torch.from_numpy(idx))
        st = torch.cuda.current_stream(self.dev)
        self.idx_dev[:B].copy_(self.pin[i][:B], non_blocking=True)
        self.pin_evt[i].record(st)
        self.pin_used[i] = True
        if counter is None:
            counter = self.counter
            self.counter += 1
        self._launch(self.idx_dev.data_ptr(), B, T, K, gap_mode, counter, self.feed, st)
        return self._describe(self.feed, B, T, K)

    # ------------------------------------------------------------------------------------ a whole schedule, resident
    def plan(self, batches) -> None:
        """Upload the example indices of a whole schedule of batches (an epoch after the bucketed shuffle, sampler.py:40-49)
        ONCE: afterwards a training loop moves nothing over PCIe — `planned()` forms every feed from HBM-resident data."""
        lens = [int(len(b)) for b in batches]
        self.plan_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        flat = np.concatenate([np.asarray(b, dtype=np.int32) for b in batches]) if batches else np.zeros(0, np.int32)
        self.plan_dev = torch.tensor(flat, device=self.dev)
        self.plan_B = lens
        self.plan_T = [int(self.in_len[int(b[0])]) for b in batches]
        # the feed buffers of planned() are sized HERE, with the schedule (outside a caller's timed loop): a longer length bucket
        # first met inside the loop must not re-allocate there (a 20-step bench loop whose 5 warm-up steps saw only T <= 2 paid
        # ~0.7 ms for it: profiles/r04_ab_experiments.txt)
        if lens:
            self._ensure_feeds(getattr(self, "_k_hint", 32))

    def _feed_need(self, K: int) -> int:
        need = max(7 * b * t + 3 * b + b * K for b, t in zip(self.plan_B, self.plan_T))
        return (need + 63) // 64 * 64

    def _ensure_feeds(self, K: int) -> int:
        """two chunk buffers for the current plan at K negatives (grown, never shrunk); the side stream and its events are made once"""
        need = self._feed_need(K)
        ch = max(1, int(self.CHUNK))
        if getattr(self, "_side", None) is None:
            # (diagnostic, round 6: TCAR_SAMPLER_STREAM=third | aux forms the chunks on one of the engine's own side streams instead of
            #  a stream of the sampler's — one stream less for HIP to map onto its hardware queues)
            which = os.environ.get("TCAR_SAMPLER_STREAM", "")
            borrowed = None
            if which in ("third", "aux"):
                try:
                    self.eng._ctx()
                    borrowed = getattr(self.eng, "_aux3" if which == "third" else "_aux", None)
                except Exception:
                    borrowed = None
            self._side = borrowed if borrowed is not None else torch.cuda.Stream(self.dev)
            self._ev_ready = [torch.cuda.Event(), torch.cuda.Event()]
            self._ev_free = [torch.cuda.Event(), torch.cuda.Event()]
        if getattr(self, "_feeds", None) is None or self._feeds[0].numel() < need * ch:
            if getattr(self, "_feeds", None) is not None:
                torch.cuda.current_stream(self.dev).synchronize()      # (steps still reading the old buffers)
                self._side.synchronize()
            self._feeds = [torch.empty(need * ch, dtype=torch.int32, device=self.dev) for _ in range(2)]
        self._k_hint = K
        return need

    # batches formed per hand-off between the side stream and the consumer's stream (planned()); TCAR_FEED_CHUNK overrides (A/B)
    CHUNK = int(os.environ.get("TCAR_FEED_CHUNK", "16"))

    @staticmethod
    def chunk_bounds(n: int, ch: int):
        """[lo, hi) of the chunks the planned batches are formed in: 2, 2, 4, 8, ... doubling up to `ch` batches (see planned)"""
        lo, size = [0], min(2, ch)
        while lo[-1] + size < n:
            lo.append(lo[-1] + size)
            if len(lo) > 2:
                size = min(2 * size, ch)
        return lo, lo[1:] + [n]

    def planned(self, K: int, gap_mode: str = "active_t"):
        """Yield the C batch descriptor of every planned batch.  The batches are formed (tcar_form_batch: session rows and
        negatives) on a side stream in CHUNKS of `CHUNK` batches, one chunk ahead of the consumer: while the consumer's steps of
        chunk j run, the feeds of chunk j + 1 are formed; two chunk buffers alternate, ONE event pair per chunk orders their
        reuse.  (Round 3 handed over every single batch: a wait and a record on the consumer's stream per step — two barrier
        packets, ~12-14 us of every 0.55-ms step, MI355X event costs in profiles/r03_event_cost.txt.)  The consumer must enqueue
        its step on the current stream before asking for the next batch."""
        n = len(self.plan_B)
        if n == 0:
            return
        need = self._ensure_feeds(K)
        ch = max(1, int(self.CHUNK))
        main = torch.cuda.current_stream(self.dev)
        side, used = self._side, [False, False]
        side.wait_stream(main)               # the plan's upload (and whatever wrote the stores) is on the main stream
        # chunk boundaries: SHORT first chunks that double up to the full size (2, 2, 4, 8, 16, 16, ...).  Forming a chunk is two
        # launches per batch on the host (~0.25 ms for 16 batches): the consumer's first step waits for the whole first chunk, and the
        # host can only afford the launches of a chunk once it is that far ahead of the device — it gains ~0.1 ms per step.  (Round 4
        # formed 2 then 16 batches BEFORE the first step was enqueued: the device idled ~0.25 ms at the top of every loop, 12 us per
        # step of a 20-step run — profiles/r05_ab_experiments.txt, tools/short_form_trace.sh.)
        lo, hi = self.chunk_bounds(n, ch)
        nchunk = len(lo)

        def launch(j):
            slot = j & 1
            if used[slot]:
                side.wait_event(self._ev_free[slot])      # the steps that read this chunk buffer have been enqueued in full
            for i in range(lo[j], hi[j]):
                self._launch(self.plan_dev.data_ptr() + 4 * int(self.plan_off[i]), self.plan_B[i], self.plan_T[i], K, gap_mode,
                             self.counter, self._feeds[slot][(i - lo[j]) * need:], side)
                self.counter += 1
            self._ev_ready[slot].record(side)

        launch(0)
        for j in range(nchunk):
            slot = j & 1
            main.wait_event(self._ev_ready[slot])
            for i in range(lo[j], hi[j]):
                yield self._describe(self._feeds[slot][(i - lo[j]) * need:], self.plan_B[i], self.plan_T[i], K)
                if i == lo[j] and j + 1 < nchunk:
                    # the next chunk is formed once the consumer has enqueued the FIRST step of this one: the device has work while
                    # the host spends the launches (the other buffer is free: its steps were enqueued before this chunk's wait)
                    launch(j + 1)
            self._ev_free[slot].record(main)
            used[slot] = True

    def read_back(self, bt: Batch) -> Dict[str, np.ndarray]:
        """The feed of `bt` as the host arrays of SessionStore.batch_arrays (+ "neg"): tests / debugging."""
        B, T, K = bt.B, bt.T, bt.K
        n = B * T
        f = self.feed[:7 * n + 3 * B + B * K].cpu().numpy()
        out = {"seq": f[:n].reshape(B, T)}
        for j, name in enumerate(("pm", "pd", "pw", "ph", "pmi", "gap")):
            out[name] = f[(j + 1) * n:(j + 2) * n].reshape(B, T)
        out["cw"], out["ch"], out["label"] = f[7 * n:7 * n + B], f[7 * n + B:7 * n + 2 * B], f[7 * n + 2 * B:7 * n + 3 * B]
        out["neg"] = f[7 * n + 3 * B:].reshape(B, K) if K else None
        return out
