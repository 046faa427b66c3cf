"""Catalog-sharded data-parallel TCAR step ("catalog-sharded scoring"): one process per GPU,
torch.distributed ("nccl" = RCCL over xGMI).  No reference equivalent — the reference is single process.

The replica exchange of dp.py moves the dense item-table gradient and the candidate-time block of dE through an all-reduce
(106 MB per step and rank at the Globo size).  Here rank r OWNS the catalog rows [n0_r, n0_r + S) of the candidate side —
their bf16 planes, their gradient, their Adam moments — and scores them against the sessions of EVERY rank:

  session forward (local B sessions)   gather, projections, pools, output transforms           -> attout [B, ek]
  all-gather   ONE packed row per session: [attout | label | negative-term coefficient | negatives]    -> [W*B, ld]
  scoring      logits = attout_all . E_shard^T  [W*B, S];  per-shard softmax statistics
  all-gather   (max, sum exp, label logit) per session and shard -> lse, cross entropy          (3 floats per session)
  gradients    dlogits planes;  dE_shard = dlogits^T attout_all (item block | time block) STAYS LOCAL;
               dX_partial = dlogits . E_shard [W*B, ek]
  reduce-scatter dX_partial over the ranks                                                     -> dattout of the local sessions
  session backward (local)            ... -> (id, row) item-row gradients of the local sessions
  all-gather   packed [row | id] item-row gradients; every owner keeps the rows of its shard
  all-reduce   arena gradients + IndexedSlices norm pieces (+ the shards' dense item norms)     (5 MB)
  update       clip + Adam: arena on every rank (identical inputs, dense norms summed in a fixed order: identical bits,
               no broadcast), item rows by their owner
  all-gather   the updated item rows [S, ldh] -> every rank's E (the session-side gathers of the next step read any row)

Per step and rank at W = 8, B = 512, N = 46,033: ~14 MB attout + ~14 MB dX + ~9 MB rows + 5 MB arena + 47 MB item rows
received, against 2 x 106 MB through the replica all-reduce; the candidate-side memory (planes, gradient, moments: 0.5 GB
at this size, 130 GB at 10 M items) divides by W.  The clip semantics of DESIGN.md S5 hold exactly: the dense item norm is
the sum of the shards' norms of (scoring + densified negative part), taken BEFORE the gathered rows are scattered in.
Six collectives per step (eleven before the exchange buffers were packed): at ~20-40 us of latency each over RCCL they,
not the bytes, are what a step waits for at this catalog size.

Sequenced from C++ between the exchanges (csrc/step.hip: tcar_step_session_forward / tcar_shard_score / tcar_shard_backward /
tcar_shard_finish / tcar_step_session_backward), the collectives in between from here; split-bf16 scoring modes only.  Evaluation scores the local sessions against the whole catalog on the fp32 GEMM.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, dp
from ._lib import Batch, Dims, GemmDesc, Segments, check
from .engine import SLOT, TcarEngine, _ru


def shard_rows(n_items: int, world: int):
    """rows per shard (a multiple of 128: the bf16 planes are blocked in 128-row units); the last shard may be short"""
    return _ru((n_items + world - 1) // world, 128)


class ShardedEngine(TcarEngine):
    def __init__(self, params, content_emb, mwdhm, lr=1e-3, max_grad=150.0, neg_weight=0.01, device="cuda:0", group=None,
                 scoring="bf16x3", world: Optional[int] = None, rank: Optional[int] = None, **kw):
        if scoring == "f32":
            raise ValueError("the catalog-sharded step runs the split-bf16 scoring modes (use mode='replica' for f32)")
        self.group = group
        live = dist.is_available() and dist.is_initialized()
        self.world = world if world is not None else (dist.get_world_size(group) if live else 1)
        self.dp_rank = rank if rank is not None else (dist.get_rank(group) if live else 0)
        # TCAR_SIM_WORLD=W on ONE process without a process group (tools, bench.py with TCAR_FORCE_DP=1): the shapes of rank 0
        # of a W-rank job — every "all-gather" repeats the local rows W times, the reduce-scatter keeps the first slice — to
        # time the per-rank compute of a large job on one GPU.  Results are meaningless beyond their shapes.
        self._sim = False
        if not live and world is None and int(os.environ.get("TCAR_SIM_WORLD", "0")) > 1:
            self.world, self.dp_rank, self._sim = int(os.environ["TCAR_SIM_WORLD"]), 0, True
        self._mwdhm_full = np.asarray(mwdhm)
        N = content_emb.shape[0] - 1
        self.S = shard_rows(N, self.world)
        n0 = min(N, self.dp_rank * self.S)
        nl = max(0, min(N, n0 + self.S) - n0)
        if nl <= 0:
            raise ValueError("more ranks than 128-row catalog blocks")
        # dX of a shard contracts over N / W catalog rows for W * B sessions: W times more output tiles and a W times shorter
        # K than the single-rank GEMM, so the split that fills the chip (and the slab bytes it writes) shrinks with W
        if "splitk" not in kw and "TCAR_SPLITK" not in os.environ:
            kw["splitk"] = max(2, -(-36 // self.world))
        super().__init__(params, content_emb, mwdhm, lr=lr, max_grad=max_grad, neg_weight=neg_weight, device=device,
                         scoring=scoring, shard=(n0, nl), **kw)
        self.n0, self.nl = n0, nl
        self.nlpad = _ru(nl, 128)
        self.n_local_items = nl
        self.backend = dist.get_backend(group) if live and self.world > 1 else "none"
        # dX is reduce-scattered where the backend can (RCCL); gloo (CPU tests, single-GPU dry runs) all-reduces it
        self.use_reduce_scatter = self.backend == "nccl"
        self.cap = 0
        self._stage = torch.zeros(self.world, self.S, self.geo.ldh, dtype=torch.float32, device=self.dev)
        self.bytes_moved = {}

    # ------------------------------------------------------------------------------------------ collectives
    def _allgather(self, t: torch.Tensor, key: str) -> torch.Tensor:
        """[..] -> [W, ..] (rank-major); world 1: a view"""
        if self.world == 1:
            return t.unsqueeze(0)
        if self._sim:
            return t.unsqueeze(0).expand((self.world,) + tuple(t.shape)).contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out.view(-1), t.reshape(-1).contiguous(), group=self.group)
        self.bytes_moved[key] = out.numel() * out.element_size()
        return out

    def _reduce_scatter_rows(self, full: torch.Tensor, cap: int, key: str) -> torch.Tensor:
        """sum over the ranks of full [W*cap, C]; returns this rank's rows [cap, C]"""
        if self.world == 1 or self._sim:
            return full[:cap]
        self.bytes_moved[key] = full.numel() * full.element_size()
        if self.use_reduce_scatter:
            out = torch.empty(cap, full.shape[1], dtype=full.dtype, device=full.device)
            dist.reduce_scatter_tensor(out, full, group=self.group)
            return out
        dist.all_reduce(full, group=self.group)              # gloo (single-GPU dry runs) has no reduce-scatter
        return full[self.dp_rank * cap:(self.dp_rank + 1) * cap]

    def exchange_info(self) -> Dict[str, object]:
        g = self.geo
        return {"mode": "sharded", "world": self.world, "shard_rows": self.S,
                "bytes_per_step": dict(self.bytes_moved),
                "item_rows_allgather_bytes": 4 * self.world * self.S * g.ldh,
                "replica_mode_allreduce_bytes": 4 * (g.N * (g.ldh + g.pt) + self.arena_n + _lib.NSLOT),
                "collectives_per_step": 6,
                "note": "all-gather [attout | label | negatives | coefficient] rows, all-gather softmax stats, reduce-scatter dX, "
                        "all-gather [row | id] item-row gradients, all-reduce arena, all-gather updated item rows; the dense item "
                        "gradient and the candidate-time block stay local"}

    # -------------------------------------------------------------------------------------------- workspace
    def _ensure_score(self, cap: int, K: int):
        g = self.geo
        if cap > self.cap or K > getattr(self, "_kcap", 0):
            cap = max(cap, self.cap)
            Bq = self.world * cap
            f32 = dict(dtype=torch.float32, device=self.dev)
            bf = dict(dtype=torch.bfloat16, device=self.dev)
            Bp = _ru(Bq, 128)
            kc = max(K, getattr(self, "_kcap", 0), 1)
            self.ld_head = _ru(g.ek + 2 + kc, 4)
            self.head_loc = torch.zeros(cap, self.ld_head, **f32)          # [attout | label | neg coefficient | negatives | pad]
            self.s_lab = torch.full((Bq,), -1, dtype=torch.int32, device=self.dev)      # unpacked by tcar_shard_score
            self.s_neg = torch.full((Bq, kc), -1, dtype=torch.int32, device=self.dev)
            self.s_coef = torch.zeros(Bq, **f32)
            self.s_logits = torch.empty(Bq, self.nlpad, **f32)
            self.s_stats = torch.empty(Bq, 3, **f32)
            self.s_lse, self.s_ce = torch.empty(Bq, **f32), torch.empty(Bq, **f32)
            self.s_a16h, self.s_a16l = torch.zeros(Bp, g.ek, **bf), torch.zeros(Bp, g.ek, **bf)
            self.s_ap16h, self.s_ap16l = torch.zeros(Bp, g.ldh + g.pt, **bf), torch.zeros(Bp, g.ldh + g.pt, **bf)
            self.s_dl16h, self.s_dl16l = torch.zeros(Bp, self.nlpad, **bf), torch.zeros(Bp, self.nlpad, **bf)
            self.s_slabs = torch.empty(self.splitk, Bq, g.ek, **f32)
            self.s_dx = torch.empty(Bq, g.ek, **f32)
            self.cap, self._kcap = cap, kc

    # two composite spans instead of the three GEMMs of the fused step: the C++ pieces between the exchanges
    TIMED_KERNELS = ("shard_score", "shard_backward")

    # ------------------------------------------------------------- timing of the scoring pieces (bench.py)
    def enable_native_timing(self, n: int):
        self._tm = {"n": n, "ev": [[], [], []]}

    def _tick3(self, kind):
        tm = getattr(self, "_tm", None)
        if tm is None or len(tm["ev"][kind]) >= tm["n"]:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(self.dev))
        tm["ev"][kind].append((e0, e1))
        return e1

    def _tock3(self, e1):
        if e1 is not None:
            e1.record(torch.cuda.current_stream(self.dev))

    def native_timing_ms(self, kind: int = 0):
        tm = getattr(self, "_tm", None)
        return [a.elapsed_time(b) for a, b in tm["ev"][kind]] if tm else []

    # ----------------------------------------------------------------------------------------------- step
    def _step(self, bt: Optional[Batch], cap: int, K: int, update: bool, T: int):
        """one training step; bt = None: a rank whose shard of the global batch is empty still joins every collective (T is
        the step's input length: the row buffers of the exchanges have the same shape on every rank)"""
        g, lib, p = self.geo, self.lib, self._p
        W, n0, nl, nlpad = self.world, self.n0, self.nl, self.nlpad
        B = bt.B if bt is not None else 0
        cap = max(cap, B, 1)
        Bq = W * cap
        self._ensure_work(max(B, 1), T)
        self._ensure_score(cap, K)
        has_neg = K > 0
        st = self._stream()
        # ---- zero the arena, session forward, and ONE packed row per session for the exchange:
        # [attout (ek) | label | coefficient of the negative term | K negatives | pad], ints as bits, row stride ld_head —
        # one all-gather instead of four, packed / unpacked by one kernel each (tcar_shard_begin / tcar_shard_score)
        ctx, sctx = self._ctx(), self._shard_ctx()
        sh = self._shard_desc(cap)
        ldh_ = self.ld_head
        head = self.head_loc[:cap]
        refresh = int(self._time_dirty)
        check(lib.tcar_shard_begin(C.byref(ctx), C.byref(bt) if bt is not None else None, cap, self._kcap, p(head), ldh_, refresh,
                                   nl, st), "tcar_shard_begin")
        head_all = self._allgather(head, "attout+labels+negatives").view(Bq, ldh_)
        sh.att_all, sh.ld_att, sh.head_K = head_all.data_ptr(), ldh_, (K if has_neg else 0)
        # ---- scoring of the shard against every session; statistics exchange; gradients (dE stays here, dX goes home)
        tk = self._tick3(0)
        check(lib.tcar_shard_score(C.byref(sctx), C.byref(sh), refresh, st), "tcar_shard_score")
        self._tock3(tk)
        self._time_dirty = False
        stats_all = self._allgather(self.s_stats[:Bq], "softmax_stats")
        tk = self._tick3(1)
        check(lib.tcar_shard_backward(C.byref(sctx), C.byref(sh), p(stats_all), st), "tcar_shard_backward")
        self._tock3(tk)
        # negative rows, shard norm, candidate-time backward: on the aux stream behind dE, beside the dX exchange and the
        # session backward (tcar_shard_join orders the main stream behind them)
        check(lib.tcar_shard_finish(C.byref(sctx), C.byref(sh), K if has_neg else 0, p(self.s_neg) if has_neg else None,
                                    p(self.s_coef) if has_neg else None, st), "tcar_shard_finish")
        dx_rows = self._reduce_scatter_rows(self.s_dx[:Bq], cap, "dX")
        # ---- session backward (local) and the sparse-row exchange: packed rows [row (ldh) | id | pad], one all-gather
        nr = cap * T
        ldr = g.ldh + 4
        if getattr(self, "_rows_cap", 0) < nr:
            self._rows_buf = torch.zeros(nr, ldr, dtype=torch.float32, device=self.dev)
            self._rows_cap = nr
        rows = self._rows_buf[:nr]
        if bt is not None:
            if not dx_rows.is_contiguous():
                dx_rows = dx_rows.contiguous()
            ce_rows = self.s_ce[self.dp_rank * cap:self.dp_rank * cap + B]
            check(lib.tcar_step_session_backward(C.byref(ctx), C.byref(bt), p(dx_rows), p(rows), ldr, nr, p(ce_rows), st),
                  "tcar_step_session_backward")
        else:
            rows.zero_()                            # an empty rank contributes padding rows only (id 0, zero payload)
        all_rows = self._allgather(rows, "rows+ids").view(-1, ldr)
        check(lib.tcar_shard_join(C.byref(ctx), st), "tcar_shard_join")
        # ids are 1-based: the rows of this shard become 1 .. nl, the rest (and the id-0 padding) fall out
        check(lib.tcar_scatter_add_rows_packed(C.byref(self.dims_cand), p(all_rows), ldr, all_rows.shape[0], n0, p(self.Gi), st),
              "tcar_scatter_add_rows_packed")
        # ---- arena exchange (gradients + norm pieces incl. the shards' dense item norms), dense-weight norms, update.  The
        # dense-weight norms are summed in a fixed order (tcar_sqnorm, one workgroup per variable): identical gradients give
        # identical norms on every rank, the replicas stay bit-identical without a broadcast.
        if W > 1 and not self._sim:
            dist.all_reduce(self.Gx, group=self.group)
            self.bytes_moved["arena"] = self.Gx.numel() * 4
        check(lib.tcar_sqnorm(p(self.G), C.byref(self.segs_dense), p(self.sqn_dense), st), "tcar_sqnorm")
        if update:
            self._update_and_share()

    def _shard_desc(self, cap: int) -> "_lib.Shard":
        key = (cap, self.s_logits.data_ptr())
        if getattr(self, "_sh_key", None) != key:
            sh = _lib.Shard()
            sh.world, sh.cap, sh.n0, sh.n_loc = self.world, cap, self.n0, self.nl
            for n, t in (("logits", self.s_logits), ("stats", self.s_stats), ("lse", self.s_lse), ("ce", self.s_ce),
                         ("a16h", self.s_a16h), ("a16l", self.s_a16l), ("ap16h", self.s_ap16h), ("ap16l", self.s_ap16l),
                         ("dl16h", self.s_dl16h), ("dl16l", self.s_dl16l), ("slabs", self.s_slabs), ("dx", self.s_dx),
                         ("lab_all", self.s_lab), ("neg_all", self.s_neg), ("coef_all", self.s_coef)):
                setattr(sh, n, t.data_ptr())
            self._sh, self._sh_key = sh, key
        return self._sh

    def _shard_ctx(self):
        """tcar_ctx_t whose candidate side is this rank's shard (tcar_step_update: arena + the owned item rows + planes)"""
        c = self._ctx()
        if getattr(self, "_sctx_src", None) is not c:
            s = _lib.Ctx()
            C.memmove(C.byref(s), C.byref(c), C.sizeof(_lib.Ctx))
            s.d = self.dims_cand
            s.E = self.E.data_ptr() + 4 * self.n0 * self.geo.ek
            self._sctx, self._sctx_src = s, c
        return self._sctx

    def _update_and_share(self):
        g = self.geo
        check(self.lib.tcar_step_update(C.byref(self._shard_ctx()), self._lr_t(), self._stream()), "tcar_step_update")
        self._after_update()
        if self.world > 1 and not self._sim:
            self._stage[self.dp_rank, :self.nl].copy_(self.E[self.n0:self.n0 + self.nl, :g.ldh])
            dist.all_gather_into_tensor(self._stage.view(-1), self._stage[self.dp_rank].reshape(-1).clone(), group=self.group)
            self.bytes_moved["item_rows"] = self._stage.numel() * 4
            self.E[:g.N, :g.ldh].copy_(self._stage.view(-1, g.ldh)[:g.N])

    # ------------------------------------------------------------------------------------------- public API
    score_batch = property(lambda self: self.world * max(self.cap, 1))

    def train_step(self, batch, bt: Optional[Batch] = None, cap_rows: Optional[int] = None, cap: Optional[int] = None,
                   T: Optional[int] = None, K: Optional[int] = None, defer_update: bool = False):
        """`batch` / `bt` = this rank's sessions (None: its shard of the global batch is empty — it still joins every
        collective and must be told the step's input length T and negative count K, which every rank knows from the bucket
        schedule).  cap = rows every rank contributes to the all-gathers (>= the largest local batch of the step; default: the
        local batch size — weak-scaling runs with equal batches); cap_rows (dp.DPEngine's argument: cap * T) is accepted
        for drop-in use.  No metadata collective, no host synchronisation."""
        # defer_update (TcarEngine's split update) is accepted for drop-in use and ignored: the sharded update is followed by
        # the all-gather of the updated rows, which the next step's gathers need — there is nothing to defer it behind
        if bt is None and batch is not None:
            bt = self.upload(batch)
        if bt is not None:
            T = bt.T
            K = bt.K if bt.neg else 0
        elif T is None or K is None:
            raise ValueError("an empty rank needs the step's T and K")
        if cap is None:
            cap = (cap_rows // max(T, 1)) if cap_rows else (bt.B if bt is not None else 1)
        self.flush()
        self._step(bt, cap, K, True, T)
        if bt is None:
            return torch.zeros(0, device=self.dev)
        return self._loss_rows(bt, cap, K)

    def _loss_rows(self, bt, cap, K):
        if K:
            return self.loss[:bt.B]
        # label_neg fed as [B, 0]: the negative term is the constant neg_weight * ln 2 (model_combine.py:142-147)
        return self.s_ce[self.dp_rank * cap:self.dp_rank * cap + bt.B] + float(np.float32(self.neg_weight) * np.float32(np.log(2.0)))

    def loss_and_grads(self, batch, bt: Optional[Batch] = None, cap_rows: Optional[int] = None, cap: Optional[int] = None):
        bt = bt or self.upload(batch)
        if cap is None:
            cap = (cap_rows // max(bt.T, 1)) if cap_rows else bt.B
        K = bt.K if bt.neg else 0
        self.flush()
        self._step(bt, cap, K, False, bt.T)
        return self._loss_rows(bt, cap, K)

    def update(self):
        self._update_and_share()

    def eval_step(self, batch, k: int = 20, bt: Optional[Batch] = None, keep_logits: bool = False):
        """local sessions against the WHOLE catalog on the fp32 GEMM (evaluation runs once per epoch)"""
        self.flush()
        bt = bt or self.upload(batch)
        g, lib, st, p = self.geo, self.lib, self._stream(), self._p
        B = bt.B
        self._ensure_work(B, bt.T)
        if not hasattr(self, "ev_logits") or self.ev_logits.shape[0] < B:
            self.ev_logits = torch.empty(max(B, self.work_B), g.Npad, dtype=torch.float32, device=self.dev)
        if k != self.topk.shape[1] or self.topk.shape[0] < B:
            self.topk = torch.empty(max(B, self.work_B), k, dtype=torch.int32, device=self.dev)
        full = Dims(g.N, g.H, g.Ht, g.ldh, g.ldt)
        if getattr(self, "_ev_mw", None) is None:
            self._ev_mw = torch.tensor(np.ascontiguousarray(self._mwdhm_full, dtype=np.int32), device=self.dev)
        check(lib.tcar_cand_time_fwd(C.byref(full), C.byref(self._time_ptrs()), p(self._ev_mw), p(self.E), st), "tcar_cand_time_fwd")
        check(lib.tcar_step_session_forward(C.byref(self._ctx()), C.byref(bt), st), "tcar_step_session_forward")
        check(lib.tcar_gemm_f32(1, B, g.N, g.ek, p(self.attout), g.ek, p(self.E), g.ek, p(self.ev_logits), g.Npad, None, 0, 0, 1,
                                st), "tcar_gemm_f32")
        check(lib.tcar_eval_rows(B, g.N, p(self.ev_logits), g.Npad, C.c_void_p(bt.label), k, p(self.rank), p(self.topk),
                                 p(self.ce), st), "tcar_eval_rows")
        out = (self.rank[:B], self.topk[:B], self.ce[:B])
        return out + (self.ev_logits[:B, :g.N].clone(),) if keep_logits else out

    # ------------------------------------------------------------------------------ inspection (tests, export)
    def _gather_shard_rows(self, local: torch.Tensor) -> np.ndarray:
        """this rank's [nl, ldh] rows of a candidate-side table -> the whole [N + 1, H] table in the reference's shape (row 0
        = the pad row, zero).  A COLLECTIVE: every rank calls it."""
        g = self.geo
        full = torch.zeros(self.world, self.S, g.ldh, dtype=torch.float32, device=self.dev)
        full[self.dp_rank, :self.nl] = local
        if self.world > 1 and not self._sim:
            dist.all_gather_into_tensor(full.view(-1), full[self.dp_rank].reshape(-1).clone(), group=self.group)
        item = np.zeros((g.N + 1, g.H), dtype=np.float32)
        item[1:] = full.view(-1, g.ldh)[:g.N, :g.H].cpu().numpy()
        return item

    def export_state(self):
        """TcarEngine.export_state for a sharded catalog: the Adam moments of the item table live on their owners, so this is
        a COLLECTIVE — every rank calls it (and gets the full state); host/model.py lets rank 0 write the file."""
        from .engine import VAR_ORDER
        self.flush()
        out = {"var/" + k: v for k, v in self.export_params().items()}         # E is whole on every rank
        for tag, arena, item in (("m/", self.M, self.Mi), ("v/", self.V, self.Vi)):
            d = self._unpack_arena(arena.cpu().numpy())
            d["item_emb"] = self._gather_shard_rows(item)
            out.update({tag + k: d[k] for k in VAR_ORDER})
        out["meta/step"] = np.asarray(self.step, dtype=np.int64)
        out["meta/beta_pow"] = np.asarray([self.b1_pow, self.b2_pow], dtype=np.float32)
        return out

    def load_state(self, st) -> None:
        """inverse of export_state: every rank reads the same file and keeps the moment rows [n0, n0 + nl) of its shard"""
        from .engine import VAR_ORDER
        self.load_params({k[4:]: np.asarray(st[k]) for k in st if k.startswith("var/")})
        g = self.geo
        if all(("m/" + k) in st and ("v/" + k) in st for k in VAR_ORDER):
            for tag, arena, item in (("m/", self.M, self.Mi), ("v/", self.V, self.Vi)):
                vals = {k: np.asarray(st[tag + k]) for k in VAR_ORDER}
                arena.copy_(torch.from_numpy(self._pack_arena(vals)))
                it = np.zeros((self.nl, g.ldh), dtype=np.float32)
                it[:, :g.H] = vals["item_emb"][1 + self.n0:1 + self.n0 + self.nl]
                item.copy_(torch.from_numpy(it))
        if "meta/step" in st:
            self.step = int(np.asarray(st["meta/step"]))
        if "meta/beta_pow" in st:
            bp = np.asarray(st["meta/beta_pow"], dtype=np.float32)
            self.b1_pow, self.b2_pow = np.float32(bp[0]), np.float32(bp[1])

    def export_grads(self):
        """summed dense gradients of the last backward, reference shapes (the item rows of every shard are all-gathered)"""
        g = self.geo
        out = self._unpack_arena(self.G.cpu().numpy())
        out["item_emb"] = self._gather_shard_rows(self.Gi)
        from collections import OrderedDict
        from .engine import VAR_ORDER
        return OrderedDict((k, out[k]) for k in VAR_ORDER)


dp._HAVE_SHARDED = True
