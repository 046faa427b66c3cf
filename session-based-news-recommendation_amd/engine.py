"""Device-resident TCAR model state and the per-batch step, driving the C-ABI of include/tcar_hip.h.

PyTorch is used here for three things only: device memory (tensors as buffers), the current HIP stream, and
(in `dp.py`) torch.distributed.  Every arithmetic operation of the step is a hand-written gfx950 kernel behind
`libtcar_hip.so`; there is no eager/CPU fallback — constructing an engine without the library or a GPU raises.

HBM layout (fp32, "padded-concat space", see include/tcar_hip.h):
  E        [Npad, ek]   candidate matrix: item table | frozen content | clipped candidate time vectors.
                        The trainable item table LIVES in E[:, 0:ldh] (no per-step concat / copy,
                        model_combine.py:135-136); Npad = N rounded up to 64, padding rows are zero.
  W/G/M/V  flat arenas  the other 22 trainable variables (padded), their gradients and Adam moments, at identical
                        offsets; the tables that receive atomic adds come first so one memset clears them.
  Gi/Mi/Vi [N, ldh]     item-table gradient and Adam moments.
  work     per-batch activations sized for the largest (B, T) seen.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import Batch, Dims, GemmDesc, Grads, Segments, Tables, check

TIME_NAMES = ["month_embedding", "day_embedding", "week_embedding", "hour_embedding", "minute_embedding"]
TIME_VOCAB = [13, 32, 8, 25, 61]
# TF creation order of the 23 trainable variables (model_combine.py:52-128) = squared-norm slot index
VAR_ORDER = ["item_emb", "dec_pos"] + TIME_NAMES + ["duration_embedding",
             "multi_attention/input_linear_trans/w_3d", "multi_attention/cont_linear_trans/w_3d",
             "multi_attention/inter_linear_trans/w_3d", "multi_attention/res_linear_trans/w_3d",
             "multi_attention/query_trans1/w1", "multi_attention/query_trans1/b1",
             "multi_attention/query_trans2/w1", "multi_attention/query_trans2/b1",
             "attout_item_cont_trans/w1", "attout_item_cont_trans/b1",
             "cont_attention/input_linear_trans/w_3d", "cont_attention/cont_linear_trans/w_3d",
             "cont_attention/res_linear_trans/w_3d", "attout_pt_trans/w1", "attout_pt_trans/b1"]
SLOT = {n: i for i, n in enumerate(VAR_ORDER)}
# Variables whose gradient is accumulated in a fixed order by the fused step of the split-bf16 modes (bitwise repeatable from
# step to step): ALL of them — the item table (sorted segmented sum, csrc/segsum.hip), the seven small tables (one workgroup
# per destination row, sources in order, embed.hip), the nine weight matrices (un-split K up to 1,536 batch rows: longer
# batches split K with float atomics), the four biases and the two residual weights (column sums in a fixed order).  In fp32
# mode the biases and residual weights still go through float atomics (ATOMIC_IN_F32).
# tests/test_gpu_configs.py::test_same_step_twice_bitwise_report keeps both lists honest.
DETERMINISTIC_GRADS: tuple = tuple(VAR_ORDER)
ATOMIC_IN_F32: tuple = tuple(n for n in VAR_ORDER if n.endswith("/b1") or n.endswith("res_linear_trans/w_3d"))


def _ru(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class Geometry:
    def __init__(self, n_items: int, H: int, Ht: int):
        self.N, self.H, self.Ht = int(n_items), int(H), int(Ht)
        self.ldh = _ru(H, 64)
        self.ldt = 64 if Ht <= 64 else (128 if Ht <= 128 else 256)
        if self.ldh > 512 or Ht > 256 or 5 * self.ldt > 512:
            raise ValueError("unsupported hidden sizes for the gfx950 kernels: H<=512, Ht<=64 (5*ldt<=512)")
        self.ic, self.pt, self.ct = 2 * self.ldh, 5 * self.ldt, 2 * self.ldt
        self.ek = self.ic + self.pt
        self.Npad = _ru(self.N, 128)        # KB32 planes are blocked in 128-row units

    def idx(self, kind: str) -> np.ndarray:
        """logical index -> padded index for a dimension of the given kind."""
        H, Ht, ldh, ldt = self.H, self.Ht, self.ldh, self.ldt
        if kind == "H":
            return np.arange(H)
        if kind == "2H":
            return np.concatenate([np.arange(H), ldh + np.arange(H)])
        if kind == "T":
            return np.arange(Ht)
        if kind == "2T":
            return np.concatenate([np.arange(Ht) + k * ldt for k in range(2)])
        if kind == "5T":
            return np.concatenate([np.arange(Ht) + k * ldt for k in range(5)])
        if kind.startswith("V"):
            return np.arange(int(kind[1:]))
        raise KeyError(kind)

    def padded(self, kind: str) -> int:
        return {"H": self.ldh, "2H": self.ic, "T": self.ldt, "2T": self.ct, "5T": self.pt}.get(kind) or int(kind[1:])


# (short name, reference variable name, row kind, col kind or None for vectors); order = arena order.
# First block = accumulated with atomics (zeroed every step); time tables + dur contiguous (tcar_grads_t).
ARENA = [
    ("pos", "dec_pos", "V40", "H"),
    ("month", "month_embedding", "V13", "T"), ("day", "day_embedding", "V32", "T"),
    ("week", "week_embedding", "V8", "T"), ("hour", "hour_embedding", "V25", "T"),
    ("minute", "minute_embedding", "V61", "T"), ("dur", "duration_embedding", "V11", "T"),
    ("m_wres", "multi_attention/res_linear_trans/w_3d", "H", None),
    ("s_wres", "cont_attention/res_linear_trans/w_3d", "H", None),
    # ---- written by GEMM / column-sum epilogues (no zeroing needed)
    ("m_win", "multi_attention/input_linear_trans/w_3d", "2H", "H"),
    ("m_wc", "multi_attention/cont_linear_trans/w_3d", "H", "H"),
    ("m_wint", "multi_attention/inter_linear_trans/w_3d", "T", "H"),
    ("q1_w", "multi_attention/query_trans1/w1", "2T", "H"), ("q1_b", "multi_attention/query_trans1/b1", "H", None),
    ("q2_w", "multi_attention/query_trans2/w1", "H", "2H"), ("q2_b", "multi_attention/query_trans2/b1", "2H", None),
    ("o_w", "attout_item_cont_trans/w1", "2H", "2H"), ("o_b", "attout_item_cont_trans/b1", "2H", None),
    ("s_win", "cont_attention/input_linear_trans/w_3d", "5T", "H"),
    ("s_wc", "cont_attention/cont_linear_trans/w_3d", "H", "H"),
    ("ot_w", "attout_pt_trans/w1", "5T", "5T"), ("ot_b", "attout_pt_trans/b1", "5T", None),
]
N_ATOMIC = 9


_HP_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def use_priority_stream(dev: torch.device) -> None:
    """Make ONE process-wide high-priority stream the current stream of `dev`.  The main chain of a step (logits ->
    softmax -> dX -> the long tail of small kernels -> Adam) is the critical path, the aux stream's dE GEMM and
    candidate-time work have slack: with the main chain on a priority -1 stream its workgroups are dispatched first
    whenever both streams have work queued (measured 0.752 -> 0.731 ms per step).  Setting the current stream once
    (instead of entering / leaving a stream per step) costs no per-step events.
    SIDE EFFECT: torch.cuda.current_stream(dev) changes for the whole process; code that mixes the engine with launches
    on the NULL stream must pass torch.cuda.current_stream().cuda_stream instead.  TCAR_NO_PRIO=1 disables it."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    hp = _HP_STREAMS.get(idx)
    if hp is None:
        hp = _HP_STREAMS[idx] = torch.cuda.Stream(dev, priority=-1)
    cur = torch.cuda.current_stream(dev)
    if cur != hp:
        hp.wait_stream(cur)
        torch.cuda.set_stream(hp)


class TcarEngine:
    def __init__(self, params: Dict[str, np.ndarray], content_emb: np.ndarray, mwdhm: np.ndarray, lr: float = 1e-3,
                 max_grad: Optional[float] = 150.0, neg_weight: float = 0.01, device: str = "cuda:0",
                 splitk: Optional[int] = None, scoring: str = "f32", shard: Optional[tuple] = None):
        """shard = (n0, n_loc): catalog-sharded data parallelism (sharded.py) — the candidate-side state (bf16 planes of E,
        dense item gradient, candidate-time block, Adam moments, inverted index) covers the catalog rows [n0, n0 + n_loc)
        only; E itself stays whole (the session-side gathers read any row)."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.TcarError("TcarEngine needs an MI355X (no CPU fallback)")
        self.dev = torch.device(device)
        self.is_cuda = self.dev.type == "cuda"
        if self.is_cuda and self.overlap and self.native and self.priority_stream and not os.environ.get("TCAR_NO_PRIO"):
            use_priority_stream(self.dev)
        N, H = content_emb.shape[0] - 1, content_emb.shape[1]
        Ht = params["month_embedding"].shape[1]
        self.geo = g = Geometry(N, H, Ht)
        self.lr, self.max_grad, self.neg_weight = float(lr), max_grad, float(neg_weight)
        self.b1, self.b2, self.eps = 0.9, 0.999, 1e-8
        self.b1_pow, self.b2_pow = np.float32(self.b1), np.float32(self.b2)
        self.step = 0
        # split-K of dX = dlogits E: 36 slabs with the 512 x 128 bf16 tile (7 N tiles x 36 = 252 workgroups), 16 in fp32
        # split-K of the dX GEMM (bf16 modes): 18 slabs of 256 x 128 tiles = 216 workgroups at B = 512 — the same grid as 36 slabs of
        # 512 x 128 tiles, half the slab bytes for the GEMM to write and the slab reduce to read back (round 4: -9 us per step at the
        # Globo, Adressa and MIND shapes; 12 / 16 / 20 / 24 measured worse, profiles/r04_ab_experiments.txt)
        # (the materialised-logits modes — bf16x3, bf16 — contract 832 columns with other tiles: 36 stays better there, 0.649 / 0.509
        #  against 0.671 / 0.517 ms per step)
        # (catalogs from 2^20 rows: 64 slabs — the dX GEMM then takes its 256 x 384 tile, 44 % fewer fill bytes per flop, and 64 slabs of
        #  [B, 672] are noise beside a contraction that long: 10 M items 13.7 -> 7.2 ms, profiles/r06_ab_experiments.txt)
        big_catalog = int(np.asarray(content_emb).shape[0]) - 1 >= (1 << 20)
        self.splitk = splitk if splitk else (16 if scoring == "f32" else (64 if big_catalog else 18) if scoring == "bf16x3-mixed" else 36)
        if os.environ.get("TCAR_SPLITK"):
            self.splitk = int(os.environ["TCAR_SPLITK"])
        # precision of the three full-catalog scoring GEMMs: "f32" (fp32 MFMA), "bf16x3" (split-bf16 planes, three
        # bf16 MFMAs per product, fp32-class accuracy), "bf16" (hi plane only)
        # "bf16x3-mixed": logits in bf16x3 (fp32-class), the two gradient GEMMs in plain bf16 (mixed-precision backward)
        if scoring not in ("f32", "bf16x3", "bf16x3-mixed", "bf16"):
            raise ValueError("scoring must be f32 | bf16x3 | bf16x3-mixed | bf16")
        self.scoring = scoring
        self.scoring_code = {"f32": 0, "bf16": 1, "bf16x3": 3, "bf16x3-mixed": 3}[scoring]
        self.scoring_bwd = 1 if scoring == "bf16x3-mixed" else 0
        f32 = dict(dtype=torch.float32, device=self.dev)
        # arena layout ------------------------------------------------------------------------------------
        self.seg = OrderedDict()
        off = 0
        for short, ref, rk, ck in ARENA:
            rows = g.padded(rk)
            cols = g.padded(ck) if ck else 1
            n = rows * cols
            self.seg[short] = dict(off=off, rows=rows, cols=cols, n=n, ref=ref, rk=rk, ck=ck, slot=SLOT[ref])
            off += n
        self.arena_n = off
        self.atomic_n = sum(self.seg[a[0]]["n"] for a in ARENA[:N_ATOMIC])
        self.W = torch.zeros(off, **f32)
        # gradients + the per-row norm pieces live in ONE buffer (a single all-reduce in the data-parallel path)
        self.Gx = torch.zeros(off + _lib.NSLOT, **f32)
        self.G = self.Gx[:off]
        self.M = torch.zeros(off, **f32)
        self.V = torch.zeros(off, **f32)
        self.E = torch.zeros(g.Npad, g.ek, **f32)
        self.shard = (0, g.N) if shard is None else (int(shard[0]), int(shard[1]))
        n0, nl = self.shard                       # candidate-side state covers rows [n0, n0 + nl)
        nlpad = _ru(nl, 128)
        if self.scoring_code:
            self.e16h = torch.zeros(nlpad, g.ek, dtype=torch.bfloat16, device=self.dev)
            self.e16l = torch.zeros(nlpad, g.ek, dtype=torch.bfloat16, device=self.dev)
        # dense item-table gradient and the candidate-side time block of dE, contiguous for the same reason
        self.big = torch.zeros(nl * (g.ldh + g.pt), **f32)
        self.Gi = self.big[:nl * g.ldh].view(nl, g.ldh)
        self.d_et = self.big[nl * g.ldh:].view(nl, g.pt)
        self.Mi = torch.zeros(nl, g.ldh, **f32)
        self.Vi = torch.zeros(nl, g.ldh, **f32)
        self.sqn_dense = torch.zeros(_lib.NSLOT, **f32)
        self.sqn_pieces = self.Gx[off:]
        use = np.ones(_lib.NSLOT, dtype=np.int32)
        for n in ["dec_pos", "duration_embedding"] + TIME_NAMES:
            use[SLOT[n]] = 0                      # tables: IndexedSlices pieces only
        self.use_dense = torch.tensor(use, device=self.dev)
        self.mwdhm = torch.tensor(np.ascontiguousarray(np.asarray(mwdhm)[n0:n0 + nl], dtype=np.int32), device=self.dev)
        self.dims = Dims(g.N, g.H, g.Ht, g.ldh, g.ldt)
        self.dims_cand = Dims(nl, g.H, g.Ht, g.ldh, g.ldt)        # the candidate-side kernels see the shard as a catalog
        # static inverted index of publish_time_MWDHM: candidates listed per time-table row (cand_time_bwd_indexed)
        mw = np.ascontiguousarray(np.asarray(mwdhm)[n0:n0 + nl], dtype=np.int64)
        rowoff = np.array([0, 13, 45, 53, 78])
        key = (np.clip(mw, 0, np.array(TIME_VOCAB) - 1) + rowoff[None, :]).T.reshape(-1)          # [5N], k-major
        order = np.argsort(key, kind="stable")
        inv_off = np.zeros(140, dtype=np.int32)
        inv_off[1:] = np.cumsum(np.bincount(key, minlength=139))
        self.inv_n = torch.tensor((order % nl).astype(np.int32), device=self.dev)
        self.inv_off = torch.tensor(inv_off, device=self.dev)
        # inverse of the index: position of (k, n) in list order.  In the bf16 scoring modes the dE GEMM writes the time
        # block of dE in THAT order (tcar_gemm_bf16_perm), so the candidate-time backward streams contiguous lists
        et_perm = np.empty(5 * nl, dtype=np.int32)
        et_perm[order] = np.arange(5 * nl, dtype=np.int32)
        self.et_perm = torch.tensor(et_perm, device=self.dev)
        self.adam_bitmap = torch.zeros(((nl + 31) // 32 + 15) // 16 * 16, dtype=torch.int32, device=self.dev)   # split update marks, whole 64-byte units
        self.ct_ws = torch.zeros(self.lib.tcar_cand_time_ws_floats(C.byref(self.dims_cand)), **f32)
        # segment tables for the optimizer kernels
        self.segs_all = self._segments([a[0] for a in ARENA])
        self.segs_dense = self._segments([a[0] for a in ARENA if self.use_dense_np(a[1])])
        self._use_np = use
        self.load_params(params, content_emb)
        self.work_rows = 0
        self.work_B = 0
        self._time_dirty = True
        self.pin = None

    def use_dense_np(self, ref: str) -> bool:
        return ref not in (["dec_pos", "duration_embedding"] + TIME_NAMES)

    def _segments(self, names) -> Segments:
        s = Segments()
        s.nseg = len(names)
        for i, n in enumerate(names):
            sg = self.seg[n]
            s.off[i], s.len[i], s.slot[i] = sg["off"], sg["n"], sg["slot"]
        return s

    # --------------------------------------------------------------------------------- parameter (un)packing
    def load_params(self, params: Dict[str, np.ndarray], content_emb: Optional[np.ndarray] = None):
        if getattr(self, "_pending_lr", None) is not None:
            self.flush()
        g = self.geo
        self.W.copy_(torch.from_numpy(self._pack_arena(params)))
        if content_emb is not None:
            self._content = np.asarray(content_emb, dtype=np.float32)
        item = np.asarray(params["item_emb"], dtype=np.float32)
        # the candidate matrix is assembled on the device in row chunks: no [Npad, ek] host image (33 GB at 10 M items)
        self.E.zero_()
        for lo in range(0, g.N, 1 << 20):
            hi = min(g.N, lo + (1 << 20))
            self.E[lo:hi, :g.H].copy_(torch.from_numpy(np.ascontiguousarray(item[1 + lo:1 + hi])))
            self.E[lo:hi, g.ldh:g.ldh + g.H].copy_(torch.from_numpy(np.ascontiguousarray(self._content[1 + lo:1 + hi])))
        if self.scoring_code:
            n0, nl = self.shard
            check(self.lib.tcar_split_bf16(self._p(self.E, n0 * g.ek), g.ek, nl, g.ek, self._p(self.e16h), self._p(self.e16l),
                                           g.ek, None, None, 0, 0, 0, self._stream()), "tcar_split_bf16")
        self._item_row0 = np.asarray(params["item_emb"], dtype=np.float32)[0].copy()
        self._time_dirty = True

    def _pack_arena(self, values: Dict[str, np.ndarray]) -> np.ndarray:
        """reference-shaped arrays of the 22 arena variables -> one padded flat fp32 arena (pads zero)"""
        g = self.geo
        W = np.zeros(self.arena_n, dtype=np.float32)
        for short, sg in self.seg.items():
            src = np.asarray(values[sg["ref"]], dtype=np.float32)
            dst = W[sg["off"]:sg["off"] + sg["n"]].reshape(sg["rows"], sg["cols"])
            ri = g.idx(sg["rk"])
            if sg["ck"] is None:
                dst[ri, 0] = src.reshape(-1)
            else:
                dst[np.ix_(ri, g.idx(sg["ck"]))] = src
        return W

    # ------------------------------------------------------------------------------------- checkpoint state
    def _probe_flag_forks(self) -> bool:
        """tcar_flag_fork_selftest: a polling kernel on the aux stream and a kernel enqueued behind it on the main stream"""
        ok = C.c_int32(0)
        check(self.lib.tcar_flag_fork_selftest(self._sig.data_ptr(), self._stream(), self._aux.cuda_stream, C.byref(ok)),
              "tcar_flag_fork_selftest")
        return bool(ok.value)

    _FORK_MSG = ("%d flag-fork poll(s) timed out: the step's streams are not running concurrently and a consumer may have run "
                 "ahead of its producer — results since the last check are invalid (set TCAR_FLAG_FORK=0 to fork with events)")

    def poll_fork_errors(self):
        """Fail fast, WITHOUT synchronising: a poll that gives up also counts in a pinned host word (tcar_ctx_t.sig_err_host,
        system-scope atomic), which the host reads here after every step.  A time-out is seen at most a step or two after it
        happened (the host runs ahead of the device by that much), long before an epoch ends or anything is reported."""
        eh = getattr(self, "_sig_err_np", None)
        if eh is not None and eh[0]:
            raise RuntimeError(self._FORK_MSG % int(eh[0]))

    def check_forks(self):
        """Raise if a polling kernel of a flag fork (tcar_ctx_t.sig_dev) ever gave up waiting: the side streams did not run
        beside the main stream (kernels serialised by a profiler's counter collection, or streams sharing one hardware queue),
        and a consumer may have run ahead of its producer.  Synchronises the device; called wherever the host reads results
        back: every evaluation step, export_state / export_params (checkpoints), the training loop's epoch end, bench.py after
        its timed loop.  Between those, poll_fork_errors() tests the host-visible mirror after every training step."""
        if getattr(self, "_sig", None) is not None:
            n = int(self._sig[32].item())
            if n:
                raise RuntimeError(self._FORK_MSG % n)
        self.poll_fork_errors()

    def export_state(self) -> Dict[str, np.ndarray]:
        """Everything a resumed run needs, as plain arrays (np.savez, loadable with allow_pickle=False): the 23 variables
        `var/<name>`, the Adam moments `m/<name>`, `v/<name>` in the reference's shapes, the beta powers and the step
        count (tf.train.AdamOptimizer's beta1_power / beta2_power non-slot variables, model_combine.py:155)."""
        self.flush()
        self.check_forks()
        g = self.geo
        out = {"var/" + k: v for k, v in self.export_params().items()}
        for tag, arena, item in (("m/", self.M, self.Mi), ("v/", self.V, self.Vi)):
            d = self._unpack_arena(arena.cpu().numpy())
            it = np.zeros((g.N + 1, g.H), dtype=np.float32)
            it[1:] = item[:, :g.H].cpu().numpy()
            d["item_emb"] = it
            out.update({tag + k: d[k] for k in VAR_ORDER})
        out["meta/step"] = np.asarray(self.step, dtype=np.int64)
        out["meta/beta_pow"] = np.asarray([self.b1_pow, self.b2_pow], dtype=np.float32)
        return out

    def load_state(self, st) -> None:
        """Inverse of export_state (moments / powers / step are optional: a variables-only file restores the weights)."""
        self.load_params({k[4:]: np.asarray(st[k]) for k in st if k.startswith("var/")})
        g = self.geo
        if all(("m/" + k) in st and ("v/" + k) in st for k in VAR_ORDER):
            for tag, arena, item in (("m/", self.M, self.Mi), ("v/", self.V, self.Vi)):
                vals = {k: np.asarray(st[tag + k]) for k in VAR_ORDER}
                arena.copy_(torch.from_numpy(self._pack_arena(vals)))
                it = np.zeros((g.N, g.ldh), dtype=np.float32)
                it[:, :g.H] = vals["item_emb"][1:]
                item.copy_(torch.from_numpy(it))
        if "meta/step" in st:
            self.step = int(np.asarray(st["meta/step"]))
        if "meta/beta_pow" in st:
            bp = np.asarray(st["meta/beta_pow"], dtype=np.float32)
            self.b1_pow, self.b2_pow = np.float32(bp[0]), np.float32(bp[1])

    def _unpack_arena(self, flat: np.ndarray) -> "OrderedDict[str, np.ndarray]":
        g = self.geo
        out = {}
        for short, sg in self.seg.items():
            src = flat[sg["off"]:sg["off"] + sg["n"]].reshape(sg["rows"], sg["cols"])
            ri = g.idx(sg["rk"])
            if sg["ck"] is None:
                v = src[ri, 0]
                out[sg["ref"]] = v.reshape(-1, 1).copy() if sg["ref"].endswith("w_3d") else v.copy()
            else:
                out[sg["ref"]] = src[np.ix_(ri, g.idx(sg["ck"]))].copy()
        return out

    def export_params(self) -> "OrderedDict[str, np.ndarray]":
        """All 23 trainable variables in the reference's shapes."""
        self.flush()
        self.check_forks()
        g = self.geo
        out = self._unpack_arena(self.W.cpu().numpy())
        item = np.zeros((g.N + 1, g.H), dtype=np.float32)
        item[0] = self._item_row0
        item[1:] = self.E[:g.N, :g.H].cpu().numpy()
        out["item_emb"] = item
        return OrderedDict((k, out[k]) for k in VAR_ORDER)

    def export_grads(self) -> "OrderedDict[str, np.ndarray]":
        """Summed dense gradients of the last backward, reference shapes (item row 0 has no gradient)."""
        g = self.geo
        out = self._unpack_arena(self.G.cpu().numpy())
        item = np.zeros((g.N + 1, g.H), dtype=np.float32)
        item[1:] = self.Gi[:, :g.H].cpu().numpy()
        out["item_emb"] = item
        return OrderedDict((k, out[k]) for k in VAR_ORDER)

    def export_sqnorms(self) -> Dict[str, float]:
        d = self.sqn_dense.cpu().numpy().astype(np.float64)
        p = self.sqn_pieces.cpu().numpy().astype(np.float64)
        return {n: float(self._use_np[i] * d[i] + p[i]) for i, n in enumerate(VAR_ORDER)}

    # ------------------------------------------------------------------------------------------- workspace
    def _ensure_work(self, B: int, T: int):
        g = self.geo
        rows = B * T
        f32 = dict(dtype=torch.float32, device=self.dev)
        if rows > self.work_rows:
            r = rows
            self.x_icp = torch.empty(r, g.ic, **f32)
            self.x_pt = torch.empty(r, g.pt, **f32)
            self.x_act = torch.empty(r, g.ldt, **f32)
            self.pre1 = torch.empty(r, g.ldh, **f32)
            self.pre2 = torch.empty(r, g.ldh, **f32)
            self.alpha = torch.empty(3 * r, **f32)
            self.dx_icp = torch.empty(r, g.ic, **f32)
            self.dx_pt = torch.empty(r, g.pt, **f32)
            self.dx_act = torch.empty(r, g.ldt, **f32)
            self.dpre1 = torch.empty(r, g.ldh, **f32)
            self.dpre2 = torch.empty(r, g.ldh, **f32)
            self.work_rows = r
        if B > self.work_B:
            self.click_t = torch.empty(B, g.ct, **f32)
            self.q1 = torch.empty(B, g.ldh, **f32)
            self.q = torch.empty(B, g.ic, **f32)
            self.pooled = torch.empty(B, g.ek, **f32)
            self.attout = torch.empty(B, g.ek, **f32)
            self.logits = torch.empty(B, g.Npad, **f32)
            self.ce = torch.empty(B, **f32)
            self.neg_fb = torch.zeros(B, **f32)
            self.loss = torch.zeros(B, **f32)
            self.neg_coef = torch.zeros(B, **f32)
            self.negpart = torch.zeros(B, g.ic, **f32)
            self.dattout = torch.empty(B, g.ek, **f32)
            self.dpooled = torch.empty(B, g.ek, **f32)
            self.dq = torch.empty(B, g.ic, **f32)
            self.dq1 = torch.empty(B, g.ldh, **f32)
            self.gw_rows = torch.empty(B, g.ic, **f32)     # per-session d w_res rows (order-fixed column sums, tcar_colsum_det)
            self.dclick = torch.empty(B, g.ct, **f32)
            self.slabs = torch.empty(self.splitk, B, g.ek, **f32)
            if self.scoring_code:
                bf = dict(dtype=torch.bfloat16, device=self.dev)
                Bp = _ru(B, 128)
                self.a16h, self.a16l = torch.zeros(Bp, g.ek, **bf), torch.zeros(Bp, g.ek, **bf)
                self.ap16h, self.ap16l = torch.zeros(Bp, g.ldh + g.pt, **bf), torch.zeros(Bp, g.ldh + g.pt, **bf)
                self.dl16h, self.dl16l = torch.zeros(Bp, g.Npad, **bf), torch.zeros(Bp, g.Npad, **bf)
            self.rank = torch.empty(B, dtype=torch.int32, device=self.dev)
            self.topk = torch.empty(B, 20, dtype=torch.int32, device=self.dev)
            self.work_B = B

    # --------------------------------------------------------------------------------------------- helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    @staticmethod
    def _p(t: torch.Tensor, off: int = 0):
        return C.c_void_p(t.data_ptr() + 4 * off)

    def _w(self, name: str):
        return C.c_void_p(self.W.data_ptr() + 4 * self.seg[name]["off"])

    def _g(self, name: str):
        return C.c_void_p(self.G.data_ptr() + 4 * self.seg[name]["off"])

    def gemm(self, layout, M, N, K, A, lda, Bm, ldb, Cm, ldc, bias=None, act=0, beta=0, splitk=1, tag=None):
        ev = self._tick(tag)
        check(self.lib.tcar_gemm_f32(layout, M, N, K, A, lda, Bm, ldb, Cm, ldc, bias, act, beta, splitk,
                                     self._stream()), "tcar_gemm_f32")
        self._tock(ev)

    @staticmethod
    def desc(M, N, segs, Cm, ldc, bias=None, act=0, beta=0, splitk=1, atomic=0) -> GemmDesc:
        """One problem of a grouped GEMM; segs = [(A, lda, B, ldb, K), ...] accumulate into one C."""
        d = GemmDesc()
        d.nseg = len(segs)
        for i, (A, lda, Bm, ldb, K) in enumerate(segs):
            d.A[i], d.lda[i], d.B[i], d.ldb[i], d.K[i] = A.value, lda, Bm.value, ldb, K
        d.C, d.ldc, d.bias = Cm.value, ldc, (bias.value if bias is not None else None)
        d.M, d.N, d.act, d.beta, d.splitk, d.atomic = M, N, act, beta, splitk, atomic
        return d

    def ggemm(self, layout, descs, tag=None):
        arr = (GemmDesc * len(descs))(*descs)
        ev = self._tick(tag)
        check(self.lib.tcar_gemm_f32_grouped(layout, len(descs), arr, self._stream()), "tcar_gemm_f32_grouped")
        self._tock(ev)

    # per-kernel device timing (bench.py): HIP events on the stream the kernels are launched on
    timing = None

    def enable_timing(self, tags):
        self.timing = {t: [] for t in tags}

    def _tick(self, tag):
        if self.timing is None or tag not in self.timing:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(self.dev))
        self.timing[tag].append((e0, e1))
        return e1

    def _tock(self, ev):
        if ev is not None:
            ev.record(torch.cuda.current_stream(self.dev))

    def timing_summary(self):
        """tag -> (launches, mean ms); call after a device synchronize."""
        out = {}
        for t, evs in (self.timing or {}).items():
            if evs:
                ms = [a.elapsed_time(b) for a, b in evs]
                out[t] = (len(ms), float(np.mean(ms)))
        return out

    def make_resident(self, batch: Dict[str, np.ndarray]) -> Batch:
        """Upload a batch into its OWN device buffer (kept alive by the engine) and return its descriptor.  The staging
        state of upload() (pinned double buffer, events, cursor) is saved and restored as a whole, so uploads before and
        after are unaffected."""
        names = ("pin", "pin_np", "ibufs", "pin_evt", "pin_used", "pin_i")
        save = {n: getattr(self, n, None) for n in names}
        self.pin = None
        try:
            bt = self.upload(batch)
            torch.cuda.current_stream(self.dev).synchronize()
            self._resident = getattr(self, "_resident", []) + [self.ibufs]
        finally:
            for n in names:
                setattr(self, n, save[n])
        return bt

    def _tables(self) -> Tables:
        t = Tables()
        t.E = self.E.data_ptr()
        t.pos = self._w("pos").value
        for k, n in enumerate(["month", "day", "week", "hour", "minute"]):
            t.time[k] = self._w(n).value
        t.dur = self._w("dur").value
        return t

    def _grads(self) -> Grads:
        gr = Grads()
        gr.g_item = self.Gi.data_ptr()
        gr.g_pos = self._g("pos").value
        for k, n in enumerate(["month", "day", "week", "hour", "minute"]):
            gr.g_time[k] = self._g(n).value
            gr.slot_time[k] = SLOT[TIME_NAMES[k]]
        gr.g_dur = self._g("dur").value
        gr.sqn = self.sqn_pieces.data_ptr()
        gr.slot_item, gr.slot_pos, gr.slot_dur = SLOT["item_emb"], SLOT["dec_pos"], SLOT["duration_embedding"]
        return gr

    def _time_ptrs(self):
        arr = (C.c_void_p * 5)()
        for k, n in enumerate(["month", "day", "week", "hour", "minute"]):
            arr[k] = self._w(n).value
        return arr

    # ----------------------------------------------------------------------------------------------- batch
    def upload(self, batch: Dict[str, np.ndarray]) -> Batch:
        """Pack the feed arrays into ONE int32 buffer, one H2D copy; returns the C batch descriptor."""
        seq = np.ascontiguousarray(batch["seq"], dtype=np.int32)
        B, T = seq.shape
        if T > 40:
            raise IndexError("session longer than the 40-row position table (model_combine.py:57)")
        if seq.min() < 1 or seq.max() > self.geo.N:
            raise IndexError("item id outside [1, N]")
        neg = batch.get("neg", None)
        K = 0 if neg is None or np.asarray(neg).size == 0 else np.asarray(neg).shape[1]
        parts = [seq.reshape(-1)] + [np.asarray(batch[k], dtype=np.int32).reshape(-1) for k in
                                     ("pm", "pd", "pw", "ph", "pmi", "gap", "cw", "ch", "label")]
        if K:
            parts.append(np.asarray(neg, dtype=np.int32).reshape(-1))
        total = sum(x.size for x in parts)
        if self.pin is None or self.pin[0].numel() < total:
            n = max(total, 1 << 16)
            self.pin = [torch.empty(n, dtype=torch.int32).pin_memory() for _ in range(2)]
            self.pin_np = [t.numpy() for t in self.pin]          # numpy views of the pinned staging buffers
            self.ibufs = [torch.empty(n, dtype=torch.int32, device=self.dev) for _ in range(2)]
            self.pin_evt = [torch.cuda.Event(), torch.cuda.Event()] if self.is_cuda else [None, None]
            self.pin_used = [False, False]
            self.pin_i = 0
        i = self.pin_i = self.pin_i ^ 1          # two staging buffers: the host may run one step ahead
        if self.pin_used[i]:
            self.pin_evt[i].synchronize()        # the H2D copy that last used this pinned buffer has finished
        np.concatenate(parts, out=self.pin_np[i][:total])        # packed straight into pinned memory
        self.ibufs[i][:total].copy_(self.pin[i][:total], non_blocking=True)
        if self.is_cuda:
            self.pin_evt[i].record(torch.cuda.current_stream(self.dev))
            self.pin_used[i] = True
        base = self.ibufs[i].data_ptr()
        bt = Batch()
        bt.B, bt.T, bt.K = B, T, K
        o = 0
        bt.seq = base
        o += B * T
        for k in range(5):
            bt.pub[k] = base + 4 * o
            o += B * T
        bt.gap = base + 4 * o
        o += B * T
        bt.cw = base + 4 * o
        o += B
        bt.ch = base + 4 * o
        o += B
        bt.label = base + 4 * o
        o += B
        bt.neg = (base + 4 * o) if K else None
        bt._seq_t = self.ibufs[i][:B * T]           # tensor view of `seq` (ids of the sparse-row exchange)
        bt._keep = self.ibufs[i]
        return bt

    # --------------------------------------------------------------------------------------------- forward
    def forward(self, bt: Batch):
        """model_combine.py:52-138 up to the full-catalog logits (Python-sequenced op-level path, fp32 scoring only;
        the bf16 scoring modes are sequenced by the C++ step driver)."""
        if self.scoring_code:
            raise _lib.TcarError("the Python-sequenced op-level path supports scoring='f32' only")
        g, lib, st = self.geo, self.lib, self._stream()
        B, T = bt.B, bt.T
        BT = B * T
        self._ensure_work(B, T)
        p = self._p
        if self._time_dirty:
            check(lib.tcar_cand_time_fwd(C.byref(self.dims), C.byref(self._time_ptrs()), p(self.mwdhm), p(self.E), st),
                  "tcar_cand_time_fwd")
            self._time_dirty = False
        tab = self._tables()
        ev = self._tick("gather_fwd")
        check(lib.tcar_gather_clip_fwd(C.byref(self.dims), C.byref(tab), C.byref(bt), p(self.x_icp), p(self.x_pt),
                                       p(self.x_act), p(self.click_t), st), "tcar_gather_clip_fwd")
        self._tock(ev)
        # one grouped launch:  pre1 = X_ic W_in + X_c W_c + X_act W_int (modules.py:126-131),
        # pre2 = X_pt W'_in + X_c W'_c (modules.py:94-96), q1 = relu(click_t Wq1 + b) (modules.py:138)
        D = self.desc
        x_c = p(self.x_icp, g.ldh)
        self.ggemm(0, [
            D(BT, g.ldh, [(p(self.x_icp), g.ic, self._w("m_win"), g.ldh, g.ic), (x_c, g.ic, self._w("m_wc"), g.ldh, g.ldh),
                          (p(self.x_act), g.ldt, self._w("m_wint"), g.ldh, g.ldt)], p(self.pre1), g.ldh),
            D(BT, g.ldh, [(p(self.x_pt), g.pt, self._w("s_win"), g.ldh, g.pt), (x_c, g.ic, self._w("s_wc"), g.ldh, g.ldh)],
              p(self.pre2), g.ldh),
            D(B, g.ldh, [(p(self.click_t), g.ct, self._w("q1_w"), g.ldh, g.ct)], p(self.q1), g.ldh,
              bias=self._w("q1_b"), act=1)])
        # q = tanh(q1 Wq2 + b)                           (modules.py:139)
        self.ggemm(0, [D(B, g.ic, [(p(self.q1), g.ldh, self._w("q2_w"), g.ic, g.ldh)], p(self.q), g.ic,
                         bias=self._w("q2_b"), act=2)])
        check(lib.tcar_attn_pool_fwd(C.byref(self.dims), B, T, p(self.x_icp), p(self.x_pt), p(self.pre1), p(self.pre2),
                                     p(self.q), self._w("m_wres"), self._w("s_wres"), p(self.pooled), p(self.alpha),
                                     st), "tcar_attn_pool_fwd")
        # attout = [tanh(pooled_ic W_o + b) | tanh(pooled_t W'_o + b)]   (model_combine.py:119,127,132)
        self.ggemm(0, [
            D(B, g.ic, [(p(self.pooled), g.ek, self._w("o_w"), g.ic, g.ic)], p(self.attout), g.ek,
              bias=self._w("o_b"), act=2),
            D(B, g.pt, [(p(self.pooled, g.ic), g.ek, self._w("ot_w"), g.pt, g.pt)], p(self.attout, g.ic), g.ek,
              bias=self._w("ot_b"), act=2)])
        # logits = attout E^T                              (model_combine.py:138)
        self.gemm(1, B, g.N, g.ek, p(self.attout), g.ek, p(self.E), g.ek, p(self.logits), g.Npad, tag="score_fwd")

    # -------------------------------------------------------------------------------------------- backward
    def backward(self, bt: Batch):
        """Loss (model_combine.py:142-147) and the gradient of its SUM w.r.t. all 23 variables."""
        self.backward_local(bt)
        lib, st, p = self.lib, self._stream(), self._p
        # clip norm of the dense item block BEFORE the sparse rows are scattered in
        self._sqnorm_item()
        tab, gr = self._tables(), self._grads()
        check(lib.tcar_gather_clip_bwd(C.byref(self.dims), C.byref(tab), C.byref(bt), p(self.dx_icp), p(self.dx_pt),
                                       p(self.dx_act), p(self.dclick), C.byref(gr), st), "tcar_gather_clip_bwd")
        self._cand_time_bwd()
        self._sqnorm_dense()

    def _sqnorm_item(self):
        g = self.geo
        one = Segments()
        one.nseg = 1
        one.off[0], one.len[0], one.slot[0] = 0, g.N * g.ldh, SLOT["item_emb"]
        check(self.lib.tcar_sqnorm(self._p(self.Gi), C.byref(one), self._p(self.sqn_dense), self._stream()), "tcar_sqnorm")

    def _cand_time_bwd(self):
        gr = self._grads()
        check(self.lib.tcar_cand_time_bwd_indexed(C.byref(self.dims), C.byref(self._time_ptrs()), self._p(self.inv_n),
                                                  self._p(self.inv_off), self._p(self.d_et), int(self.scoring_code != 0),
                                                  self._p(self.ct_ws), C.byref(gr), self._stream()),
              "tcar_cand_time_bwd_indexed")

    def _sqnorm_dense(self):
        check(self.lib.tcar_sqnorm(self._p(self.G), C.byref(self.segs_dense), self._p(self.sqn_dense), self._stream()),
              "tcar_sqnorm")

    def backward_local(self, bt: Batch):
        """Everything of the backward pass that needs no other rank: loss, dlogits, dE, input / weight gradients."""
        g, lib, st = self.geo, self.lib, self._stream()
        B, T, K = bt.B, bt.T, bt.K
        BT = B * T
        p = self._p
        self.Gx.zero_()               # tables, bias, weight gradients and norm pieces are accumulated with atomics
        self.sqn_dense.zero_()
        ev = self._tick("softmax_ce")
        check(lib.tcar_softmax_ce(B, g.N, p(self.logits), g.Npad, C.c_void_p(bt.label), p(self.ce), st), "tcar_softmax_ce")
        self._tock(ev)
        # d attout = dlogits E  (contraction over the catalog: split-K slabs + reduce)
        S = lib.tcar_gemm_splitk_effective(g.Npad, self.splitk)
        self.gemm(0, B, g.ek, g.Npad, p(self.logits), g.Npad, p(self.E), g.ek, p(self.slabs), g.ek, splitk=self.splitk,
                  tag="score_dx")
        check(lib.tcar_splitk_reduce(p(self.slabs), S, B, g.ek, g.ek, p(self.dattout), st), "tcar_splitk_reduce")
        # dE = dlogits^T attout: item columns -> Gi, time columns -> d_et (content is frozen); one launch, both
        # problems stream the same dlogits tiles
        D = self.desc
        self.ggemm(2, [
            D(g.N, g.ldh, [(p(self.logits), g.Npad, p(self.attout), g.ek, B)], p(self.Gi), g.ldh),
            D(g.N, g.pt, [(p(self.logits), g.Npad, p(self.attout, g.ic), g.ek, B)], p(self.d_et), g.pt)], tag="score_dE")
        if K:
            check(lib.tcar_neg_term(C.byref(self.dims), B, K, p(self.E), C.c_void_p(bt.neg), p(self.attout),
                                    self.neg_weight, p(self.neg_fb), p(self.dattout), p(self.Gi), p(self.ce), p(self.loss),
                                    st), "tcar_neg_term")
        else:
            self.neg_fb[:B].zero_()
        # output transforms (linear_2d + tanh) backward
        check(lib.tcar_dact_colsum(B, g.ic, g.ek, p(self.attout), p(self.dattout), self._g("o_b"), 2, st), "dact")
        check(lib.tcar_dact_colsum(B, g.pt, g.ek, p(self.attout, g.ic), p(self.dattout, g.ic), self._g("ot_b"), 2, st), "dact")
        self.ggemm(1, [
            D(B, g.ic, [(p(self.dattout), g.ek, self._w("o_w"), g.ic, g.ic)], p(self.dpooled), g.ek),
            D(B, g.pt, [(p(self.dattout, g.ic), g.ek, self._w("ot_w"), g.pt, g.pt)], p(self.dpooled, g.ic), g.ek)])
        check(lib.tcar_attn_pool_bwd(C.byref(self.dims), B, T, p(self.x_icp), p(self.x_pt), p(self.pre1), p(self.pre2),
                                     p(self.q), self._w("m_wres"), self._w("s_wres"), p(self.alpha), p(self.dpooled),
                                     p(self.dx_icp), p(self.dx_pt), p(self.dq), p(self.dpre1), p(self.dpre2),
                                     self._g("m_wres"), self._g("s_wres"), st), "tcar_attn_pool_bwd")
        # query MLP backward (modules.py:138-139)
        check(lib.tcar_dact_colsum(B, g.ic, g.ic, p(self.q), p(self.dq), self._g("q2_b"), 2, st), "dact")
        self.ggemm(1, [D(B, g.ldh, [(p(self.dq), g.ic, self._w("q2_w"), g.ic, g.ic)], p(self.dq1), g.ldh)])
        check(lib.tcar_dact_colsum(B, g.ldh, g.ldh, p(self.q1), p(self.dq1), self._g("q1_b"), 1, st), "dact")
        # input gradients: click query rows and the projections (only the ITEM half of dX_ic is needed: content
        # is frozen)
        self.ggemm(1, [
            D(B, g.ct, [(p(self.dq1), g.ldh, self._w("q1_w"), g.ldh, g.ldh)], p(self.dclick), g.ct),
            D(BT, g.ldh, [(p(self.dpre1), g.ldh, self._w("m_win"), g.ldh, g.ldh)], p(self.dx_icp), g.ic, beta=1),
            D(BT, g.ldt, [(p(self.dpre1), g.ldh, self._w("m_wint"), g.ldh, g.ldh)], p(self.dx_act), g.ldt),
            D(BT, g.pt, [(p(self.dpre2), g.ldh, self._w("s_win"), g.ldh, g.ldh)], p(self.dx_pt), g.pt, beta=1)])
        # all nine weight gradients (x^T dy, K = batch rows) in one launch, split-K with fp32 atomics into the
        # zeroed gradient arena
        kb = max(1, min(16, (B + 1023) // 1024))
        kr = max(1, min(16, (BT + 1023) // 1024))
        x_c = p(self.x_icp, g.ldh)
        W = lambda M, N, A, lda, Bm, ldb, K, name, ks: D(M, N, [(A, lda, Bm, ldb, K)], self._g(name), N, splitk=max(ks, 2),
                                                          atomic=1)
        self.ggemm(2, [
            W(g.ic, g.ic, p(self.pooled), g.ek, p(self.dattout), g.ek, B, "o_w", kb),
            W(g.pt, g.pt, p(self.pooled, g.ic), g.ek, p(self.dattout, g.ic), g.ek, B, "ot_w", kb),
            W(g.ldh, g.ic, p(self.q1), g.ldh, p(self.dq), g.ic, B, "q2_w", kb),
            W(g.ct, g.ldh, p(self.click_t), g.ct, p(self.dq1), g.ldh, B, "q1_w", kb),
            W(g.ic, g.ldh, p(self.x_icp), g.ic, p(self.dpre1), g.ldh, BT, "m_win", kr),
            W(g.ldh, g.ldh, x_c, g.ic, p(self.dpre1), g.ldh, BT, "m_wc", kr),
            W(g.ldt, g.ldh, p(self.x_act), g.ldt, p(self.dpre1), g.ldh, BT, "m_wint", kr),
            W(g.pt, g.ldh, p(self.x_pt), g.pt, p(self.dpre2), g.ldh, BT, "s_win", kr),
            W(g.ldh, g.ldh, x_c, g.ic, p(self.dpre2), g.ldh, BT, "s_wc", kr)], tag="weight_grads")

    # ---------------------------------------------------------------------------------------------- update
    def update(self):
        """model_combine.py:157-163: per-variable clip_by_norm(max_grad) + TF-1 Adam."""
        g, lib, st, p = self.geo, self.lib, self._stream(), self._p
        lr_t = self._lr_t()
        clip = float(self.max_grad) if self.max_grad else 0.0
        check(lib.tcar_clip_adam(p(self.W), p(self.G), p(self.M), p(self.V), C.byref(self.segs_all), p(self.sqn_dense),
                                 p(self.sqn_pieces), p(self.use_dense), clip, lr_t, self.b1, self.b2, self.eps, st),
              "tcar_clip_adam")
        ev = self._tick("adam_item")
        check(lib.tcar_clip_adam_2d(p(self.E), g.ek, p(self.Gi), p(self.Mi), p(self.Vi), g.N, g.ldh, SLOT["item_emb"],
                                    p(self.sqn_dense), p(self.sqn_pieces), p(self.use_dense), clip, lr_t, self.b1,
                                    self.b2, self.eps, st), "tcar_clip_adam_2d")
        self._tock(ev)
        self._after_update()

    # ------------------------------------------------------------------------------ native (C++) step driver
    overlap = os.environ.get("TCAR_NO_OVERLAP", "") == ""      # second HIP stream for the independent dE / candidate-time chains (C++ driver only)
    native = True        # drive the step from libtcar_hip.so (tcar_train_step / tcar_eval_step); False = Python

    def _ctx(self) -> "_lib.Ctx":
        """tcar_ctx_t for the current workspace (rebuilt when a buffer is re-allocated)."""
        key = (self.work_rows, self.work_B, self.topk.data_ptr() if hasattr(self, "topk") else 0, id(self._ev))
        if getattr(self, "_ctx_key", None) == key:
            return self._ctx_obj
        g, c = self.geo, _lib.Ctx()
        c.d = self.dims
        c.splitk = self.splitk
        for i, (short, sg) in enumerate(self.seg.items()):
            c.slot_of[i], c.off[i] = sg["slot"], sg["off"]
        c.slot_item = SLOT["item_emb"]
        c.b1, c.b2, c.eps = self.b1, self.b2, self.eps
        c.clip = float(self.max_grad) if self.max_grad else 0.0
        c.neg_weight = self.neg_weight
        for n, t in (("E", self.E), ("W", self.W), ("Gx", self.Gx), ("M", self.M), ("V", self.V), ("big", self.big),
                     ("Mi", self.Mi), ("Vi", self.Vi), ("sqn_dense", self.sqn_dense), ("use_dense", self.use_dense),
                     ("mwdhm", self.mwdhm), ("inv_n", self.inv_n), ("inv_off", self.inv_off), ("ct_ws", self.ct_ws),
                     ("rank", self.rank), ("topk", self.topk)):
            setattr(c, n, t.data_ptr())
        c.arena_n = self.arena_n
        c.segs_all, c.segs_dense = self.segs_all, self.segs_dense
        for n in _lib._WS:
            setattr(c, n, getattr(self, n).data_ptr())
        c.scoring = self.scoring_code
        c.et_perm = self.et_perm.data_ptr()
        c.adam_bitmap = self.adam_bitmap.data_ptr()
        c.scoring_bwd = self.scoring_bwd
        if self.scoring_code and not os.environ.get("TCAR_ATOMIC_COLSUMS"):
            c.gw_rows = self.gw_rows.data_ptr()
        wks = max(1, int((self.tune if self.tune is not None else _lib.tuning()).wgrad_ks))
        if self.work_rows > wks and not os.environ.get("TCAR_ATOMIC_WGRAD"):
            # batches of more than TCAR_WGRAD_KS (1,536) rows split the K of the weight gradients: slabs folded in split order (order-fixed)
            per = g.ic * g.ic + g.pt * g.pt + g.ldh * g.ic + g.ct * g.ldh + g.ic * g.ldh + 2 * g.ldh * g.ldh + g.ldt * g.ldh + g.pt * g.ldh
            need = min(16, (self.work_rows + wks - 1) // wks) * per
            if getattr(self, "_wgrad_slabs", None) is None or self._wgrad_slabs.numel() < need:
                self._wgrad_slabs = torch.empty(need, dtype=torch.float32, device=self.dev)
            c.wgrad_slabs, c.wgrad_slab_floats = self._wgrad_slabs.data_ptr(), self._wgrad_slabs.numel()
        if self.scoring_code:
            for n in ("e16h", "e16l", "a16h", "a16l", "ap16h", "ap16l", "dl16h", "dl16l"):
                setattr(c, n, getattr(self, n).data_ptr())
            # slab workspace of the split session-side GEMMs (forward projections: one slab per 128-deep K chunk; backward:
            # the input gradient of the output transforms)
            u = lambda k: (k + 127) // 128
            need = max((u(g.ic) + 2 * u(g.ldh) + u(g.ldt) + u(g.pt)) * self.work_rows * g.ldh,
                       max(u(g.ic), u(g.pt)) * self.work_B * g.ek)
            if getattr(self, "_proj_slabs", None) is None or self._proj_slabs.numel() < need:
                self._proj_slabs = torch.empty(need, dtype=torch.float32, device=self.dev)
            c.proj_slabs, c.proj_slab_floats = self._proj_slabs.data_ptr(), self._proj_slabs.numel()
            if self.scoring_bwd == 1:
                # softmax epilogue of the logits GEMM (training steps): per-group (max, sum) pairs, label scores, row statistics
                need = self.work_B * ((g.N + 63) // 64 + 8) * 2 + 4 * self.work_B + 8
                if getattr(self, "_ce_ws", None) is None or self._ce_ws.numel() < need:
                    self._ce_ws = torch.empty(need, dtype=torch.float32, device=self.dev)
                    self._ce_geo = (C.c_int32 * 2)(0, 0)
                c.ce_ws, c.ce_ws_floats, c.ce_geo = self._ce_ws.data_ptr(), self._ce_ws.numel(), C.cast(self._ce_geo, C.c_void_p)
                # one-hot form of the candidate-side time scores: static OH plane of publish_time_MWDHM, per-step score planes
                if self.shard == (0, g.N) and not os.environ.get("TCAR_NO_ONEHOT"):
                    if getattr(self, "_oh16", None) is None:
                        self._oh16 = torch.empty(g.Npad * 160, dtype=torch.bfloat16, device=self.dev)
                        check(self.lib.tcar_time_onehot(C.byref(self.dims), self._p(self.mwdhm), self._p(self._oh16), 160,
                                                        self._stream()), "tcar_time_onehot")
                    Bp = _ru(self.work_B, 128)
                    if getattr(self, "_p16h", None) is None or self._p16h.numel() < Bp * 160:
                        self._p16h = torch.zeros(Bp * 160, dtype=torch.bfloat16, device=self.dev)
                        self._p16l = torch.zeros(Bp * 160, dtype=torch.bfloat16, device=self.dev)
                    c.oh16, c.p16h, c.p16l = self._oh16.data_ptr(), self._p16h.data_ptr(), self._p16l.data_ptr()
                    # one-hot form of the scoring gradients: clipped time rows, dP = dlogits OH, per-candidate (q, z) pairs
                    if g.ldt == 64 and not os.environ.get("TCAR_NO_ONEHOT_BWD"):
                        if getattr(self, "_tclip", None) is None:
                            self._tclip = torch.zeros(160 * g.ldt + 320, dtype=torch.float32, device=self.dev)
                            self._qz = torch.zeros(5 * g.N * 2, dtype=torch.float32, device=self.dev)
                        if getattr(self, "_dP", None) is None or self._dP.numel() < self.work_B * 160:
                            self._dP = torch.zeros(self.work_B * 160, dtype=torch.float32, device=self.dev)
                        c.tclip, c.dP, c.qz = self._tclip.data_ptr(), self._dP.data_ptr(), self._qz.data_ptr()
                        # anchored softmax form (tcar_hip.h: ce_rowscale ...): row scales, the scaled attout plane of dE, and the
                        # host int that carries the form from the forward to the backward half of a step
                        if not os.environ.get("TCAR_NO_CE_ANCHOR"):
                            Bp = _ru(self.work_B, 128)
                            if getattr(self, "_ce_rowscale", None) is None or self._ce_rowscale.numel() < 2 * Bp:
                                self._ce_rowscale = torch.zeros(2 * Bp, dtype=torch.float32, device=self.dev)
                                self._aps16h = torch.zeros(Bp, g.ldh + g.pt, dtype=torch.bfloat16, device=self.dev)
                                self._ce_form = (C.c_int32 * 1)(0)
                            c.ce_rowscale, c.aps16h = self._ce_rowscale.data_ptr(), self._aps16h.data_ptr()
                            c.ce_form = C.cast(self._ce_form, C.c_void_p)
        if self.overlap:
            if not hasattr(self, "_aux"):
                self._aux = torch.cuda.Stream(self.dev)
                self._aux_ev = [torch.cuda.Event() for _ in range(6)]
                for e in self._aux_ev:
                    e.record(torch.cuda.current_stream(self.dev))      # materialise the hipEvent_t handles
            c.stream2 = self._aux.cuda_stream
            for i, e in enumerate(self._aux_ev):
                c.ev[i] = e.cuda_event
            # workspace of the deterministic item-row scatter (sort + segmented sum), sized for the workspace's largest batch
            need = int(self.lib.tcar_segsum_ws_bytes(C.byref(self.dims), max(1, self.work_rows + self.work_B * 64)))
            if getattr(self, "_segsum_ws", None) is None or self._segsum_ws.numel() < need:
                self._segsum_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
            c.segsum_ws, c.segsum_bytes = self._segsum_ws.data_ptr(), self._segsum_ws.numel()
            # row pieces + chunk partials of the order-fixed small-table backward (long buckets: a workgroup per ~1,024 sources)
            if getattr(self, "_small_det_ws", None) is None:
                self._small_det_ws = torch.empty(int(self.lib.tcar_small_det_ws_floats()), dtype=torch.float32, device=self.dev)
            c.small_det_ws, c.small_det_ws_floats = self._small_det_ws.data_ptr(), self._small_det_ws.numel()
            if not os.environ.get("TCAR_NO_STREAM3"):
                if not hasattr(self, "_aux3"):
                    self._aux3 = torch.cuda.Stream(self.dev)
                    self._aux3_ev = torch.cuda.Event()
                    self._aux3_ev.record(torch.cuda.current_stream(self.dev))
                c.stream3 = self._aux3.cuda_stream
                c.ev3 = self._aux3_ev.cuda_event
            # flag forks (tcar_ctx_t.sig_dev): a polling kernel must never sit in FRONT of the work it waits for in a hardware queue.
            # Every poll of the step is enqueued BEHIND its producer's launch, and a producer depends only on work enqueued before
            # it — so whatever shares the poll's queue ahead of it (another of our streams, a collective's stream) never waits for
            # the poll: the single-process engine and the catalog-sharded one (whose pieces between the collectives fork and join
            # our own streams only) use them; DPEngine, whose exchange is enqueued in the middle of the fused backward, keeps events
            if self.flag_forks and not os.environ.get("TCAR_NO_FLAG_FORK"):
                if not hasattr(self, "_sig"):
                    self._sig = torch.zeros(80, dtype=torch.int32, device=self.dev)
                    # the driver's fork slots and epoch counter: host memory owned by THIS engine (nothing per thread / process)
                    self._fork_host = (C.c_uint8 * int(self.lib.tcar_fork_state_bytes()))()
                    # time-outs are mirrored into a pinned, device-visible host word: poll_fork_errors() reads it without a sync
                    self._sig_err = torch.zeros(16, dtype=torch.int32).pin_memory()
                    self._sig_err_np = self._sig_err.numpy()
                    # one probe: do the side streams run BESIDE the main stream here?  (Not under a counter-collecting
                    # profiler or with serialised kernels: every poll would sit out its time-out — events then.)
                    if not self._probe_flag_forks():
                        import warnings
                        warnings.warn("tcar: kernels of different streams do not run concurrently here (profiler counter "
                                      "collection / serialised kernels / shared hardware queue): flag forks off, events instead")
                        self._sig = None
                if self._sig is not None:
                    c.sig_dev, c.fork_host = self._sig.data_ptr(), C.cast(self._fork_host, C.c_void_p)
                    c.sig_err_host = self._sig_err.data_ptr()
        if self.tune is not None:
            c.tune = C.cast(C.pointer(self.tune), C.c_void_p)
        if not os.environ.get("TCAR_NO_FOLD_SCRATCH"):
            # zeroed words of the order-fixed last-arrival fold (dense-weight norms of the fused step: several workgroups per variable)
            if getattr(self, "_fold_scratch", None) is None:
                self._fold_scratch = torch.zeros(128, dtype=torch.int32, device=self.dev)
            c.fold_scratch, c.fold_scratch_words = self._fold_scratch.data_ptr(), self._fold_scratch.numel()
        if self._ev is not None:
            c.ev_start = C.cast(self._ev["start_arr"], C.c_void_p)
            c.ev_stop = C.cast(self._ev["stop_arr"], C.c_void_p)
            c.ev_n = self._ev["n"]
            c.ev_cursor = C.cast(self._ev["cursor"], C.c_void_p)
        self._ctx_key, self._ctx_obj = key, c
        return c

    flag_forks = True    # may this engine class fork / join its streams through device flags (see _ctx)
    # does the engine make a process-wide high-priority stream the current one (use_priority_stream)?  The data-parallel engines
    # switch it off while collectives are live: with RCCL's copies / kernels on a priority stream AND the device sampler forming
    # batches on its normal-priority side stream, the step ran at 1.39 ms instead of 0.74 (round 6, one rank, every collective
    # forced: profiles/r06_ab_experiments.txt section 4)
    priority_stream = True
    _ev = None
    tune = None          # optional _lib.Tuning copy of THIS engine (set_tuning); None = the process-wide switch values

    def set_tuning(self, **overrides):
        """Give this engine its own copy of the TCAR_* switches with `overrides` applied (tests, tools: e.g.
        set_tuning(TCAR_FLAG_FORK=0)); other engines and the process-wide values are untouched."""
        self.flush()
        self.tune = _lib.tuning(**overrides)
        self._ctx_key = None

    TIMED_KERNELS = ("score_fwd", "score_dx", "score_dE", "session_proj", "gather_fwd")      # kinds 0..4 of tcar_ctx_t.ev_start / ev_stop

    def enable_native_timing(self, n: int):
        """HIP events around the three full-catalog GEMMs inside tcar_train_step (logits, dX, dE), each pair recorded on the
        stream its GEMM is launched on (bench.py roofline): n slots per kernel, used round-robin."""
        st = torch.cuda.current_stream(self.dev)
        k = len(self.TIMED_KERNELS)
        starts = [torch.cuda.Event(enable_timing=True) for _ in range(k * n)]
        stops = [torch.cuda.Event(enable_timing=True) for _ in range(k * n)]
        for e in starts + stops:
            e.record(st)                      # materialise the underlying hipEvent_t
        self._ev = {"n": n, "starts": starts, "stops": stops, "cursor": (C.c_int32 * 2)(0, 0),
                    "start_arr": (C.c_void_p * (k * n))(*[e.cuda_event for e in starts]),
                    "stop_arr": (C.c_void_p * (k * n))(*[e.cuda_event for e in stops])}

    def native_timing_ms(self, kind: int = 0):
        """per-launch milliseconds of kernel `kind` (index into TIMED_KERNELS); call after a device synchronize"""
        n = self._ev["n"]
        used = min(self._ev["cursor"][0], n)
        return [self._ev["starts"][kind * n + i].elapsed_time(self._ev["stops"][kind * n + i]) for i in range(used)]

    def step_form(self, bt: Batch) -> Dict[str, bool]:
        """The form a fused training step of `bt` takes on this engine (tcar_step_form: the driver's own predicates) —
        {"fused_ce", "onehot_fwd", "onehot_bwd", "sorted_rows", "ce_anchored"}.  Tools that label measurements ask this instead of
        re-deriving it from the environment."""
        self._ensure_work(bt.B, bt.T)
        out = (C.c_int32 * 5)()
        check(self.lib.tcar_step_form(C.byref(self._ctx()), C.byref(bt), out), "tcar_step_form")
        return {"fused_ce": bool(out[0]), "onehot_fwd": bool(out[1]), "onehot_bwd": bool(out[2]), "sorted_rows": bool(out[3]),
                "ce_anchored": bool(out[4])}

    def _lr_t(self) -> float:
        return float(np.float32(self.lr) * np.sqrt(np.float32(1) - self.b2_pow) / (np.float32(1) - self.b1_pow))

    def _after_update(self):
        self.step += 1
        self.b1_pow = np.float32(self.b1_pow * np.float32(self.b1))
        self.b2_pow = np.float32(self.b2_pow * np.float32(self.b2))
        self._time_dirty = True

    # ------------------------------------------------------------------------------------------ public API
    def _loss_view(self, bt: Batch) -> torch.Tensor:
        """per-session loss of model_combine.py:147 — written by the negative-term kernel.  Without negatives the reference
        still feeds label_neg as [B, 0]: neg_logits = 0 and every session's loss carries the constant
        neg_weight * -log(1 - sigmoid(0)) = neg_weight * ln 2 (no gradient)."""
        if bt.K > 0 and bt.neg:
            return self.loss[:bt.B]
        return self.ce[:bt.B] + float(np.float32(self.neg_weight) * np.float32(np.log(2.0)))

    _pending_lr = None        # bias-corrected rate of an optimizer update that has been deferred (train_step(defer_update=True))

    def flush(self):
        """Apply a deferred optimizer update now (no-op otherwise).  Every entry point except train_step(defer_update=True)
        calls it first, so a deferred update is never observable."""
        if self._pending_lr is not None:
            lr, self._pending_lr = self._pending_lr, None
            check(self.lib.tcar_step_update(C.byref(self._ctx()), lr, self._stream()), "tcar_step_update")

    def train_step(self, batch: Dict[str, np.ndarray], bt: Optional[Batch] = None, defer_update: bool = False) -> torch.Tensor:
        """One sess.run([loss, global_step, train_op]) (model_combine.py:231); returns loss[B] on device.
        defer_update=True (training loops): the Adam update of THIS step is applied at the start of the next train_step —
        arena + the item rows that step gathers first, the other item rows on the aux stream beside its forward pass
        (tcar_train_step_deferred) — or by flush(); results are identical, the HBM-bound pass over the item table leaves
        the critical path."""
        bt = bt or self.upload(batch)
        if self.native and self.timing is None:
            self._ensure_work(bt.B, bt.T)
            if defer_update or self._pending_lr is not None:
                pend, lr_p = (1, self._pending_lr) if self._pending_lr is not None else (0, 0.0)
                self._pending_lr = None
                check(self.lib.tcar_train_step_deferred(C.byref(self._ctx()), C.byref(bt), int(self._time_dirty), pend, lr_p,
                                                        self._stream()), "tcar_train_step_deferred")
                self._pending_lr = self._lr_t()
                self._after_update()              # step count / beta powers advance now; the device work is owed
                self.poll_fork_errors()
                if not defer_update:
                    self.flush()
            else:
                check(self.lib.tcar_train_step(C.byref(self._ctx()), C.byref(bt), int(self._time_dirty), self._lr_t(),
                                               self._stream()), "tcar_train_step")
                self._after_update()
                self.poll_fork_errors()
        else:
            self.flush()
            self.forward(bt)
            self.backward(bt)
            self.update()
        return self._loss_view(bt)

    def loss_and_grads(self, batch, bt: Optional[Batch] = None) -> torch.Tensor:
        self.flush()
        bt = bt or self.upload(batch)
        if self.native and self.timing is None:
            self._ensure_work(bt.B, bt.T)
            ctx, st = self._ctx(), self._stream()
            check(self.lib.tcar_step_forward(C.byref(ctx), C.byref(bt), int(self._time_dirty), st), "tcar_step_forward")
            self._time_dirty = False
            check(self.lib.tcar_step_backward_local(C.byref(ctx), C.byref(bt), st), "tcar_step_backward_local")
            check(self.lib.tcar_step_finish(C.byref(ctx), C.byref(bt), st), "tcar_step_finish")
        else:
            self.forward(bt)
            self.backward(bt)
        return self._loss_view(bt)

    # ---- diversity metrics of the evaluation loop on the device (model_combine.py:174-194,301-313)
    def set_categories(self, cat_of_item: np.ndarray):
        """cat_of_item [N]: integer category code of every 0-based item (host/metrics.category_table); also allocates the
        byte map of recommended items behind `coverage()`."""
        cat = np.ascontiguousarray(np.asarray(cat_of_item).reshape(-1), dtype=np.int32)
        if cat.shape[0] != self.geo.N:
            raise ValueError("category table must have one entry per catalog item")
        self._cat = torch.tensor(cat, device=self.dev)
        self._seen = torch.zeros(self.geo.N, dtype=torch.uint8, device=self.dev)

    def reset_coverage(self):
        self._seen.zero_()

    def eval_diversity(self, bt: Batch, topk: torch.Tensor):
        """getILD / getUnexp pair counts of the sessions of `bt` for their top-k lists (int32 [B] each, on the device) and
        the recommended items marked in the coverage map; divide with host.metrics.diversity_from_counts."""
        B, k = bt.B, int(topk.shape[1])
        topk = topk[:B].contiguous()
        out = torch.empty(3, B, dtype=torch.int32, device=self.dev)
        p = self._p
        check(self.lib.tcar_eval_diversity(B, bt.T, k, self.geo.N, p(topk), C.c_void_p(bt.seq), p(self._cat), p(out[0]), p(out[1]),
                                           p(out[2]), C.c_void_p(self._seen.data_ptr()), self._stream()), "tcar_eval_diversity")
        return out[0], out[1], out[2]

    def coverage(self) -> int:
        """number of distinct items recommended since reset_coverage() (len(resultItemDict), model_combine.py:313)"""
        return int(self._seen.sum(dtype=torch.int64).item())

    def eval_step(self, batch, k: int = 20, bt: Optional[Batch] = None, keep_logits: bool = False):
        """sess.run([softmax_input, cross_loss]) (model_combine.py:283) + rank / top-k on device.
        Returns (rank[B] int32, topk[B,k] int32, ce[B] f32[, logits [B,N]])."""
        self.flush()
        bt = bt or self.upload(batch)
        B = bt.B
        self._ensure_work(B, bt.T)
        if k != self.topk.shape[1] or self.topk.shape[0] < B:
            self.topk = torch.empty(max(B, self.work_B), k, dtype=torch.int32, device=self.dev)
        if self.native and not keep_logits:
            check(self.lib.tcar_eval_step(C.byref(self._ctx()), C.byref(bt), int(self._time_dirty), k, self._stream()),
                  "tcar_eval_step")
            self._time_dirty = False
            self.poll_fork_errors()
            return (self.rank[:B], self.topk[:B], self.ce[:B])
        if self.native:
            check(self.lib.tcar_step_forward(C.byref(self._ctx()), C.byref(bt), int(self._time_dirty), self._stream()),
                  "tcar_step_forward")
            self._time_dirty = False
        else:
            self.forward(bt)
        g, lib, st, p = self.geo, self.lib, self._stream(), self._p
        check(lib.tcar_rank_topk(B, g.N, p(self.logits), g.Npad, C.c_void_p(bt.label), k, p(self.rank), p(self.topk), st),
              "tcar_rank_topk")
        logits = self.logits[:B, :g.N].clone() if keep_logits else None
        check(lib.tcar_softmax_ce(B, g.N, p(self.logits), g.Npad, C.c_void_p(bt.label), p(self.ce), st), "tcar_softmax_ce")
        out = (self.rank[:B], self.topk[:B], self.ce[:B])
        return out + (logits,) if keep_logits else out
