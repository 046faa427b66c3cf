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


class ShardExchange:
    """The collective schedule of the catalog-sharded step, independent of where the local pieces come from (the engine binds
    them to the HIP entry points; tests/test_sharded_gloo.py drives the SAME schedule over gloo on CPU tensors with fp64
    torch pieces).  Every rank — also one whose shard of the batch is empty — issues the same six collectives in the same
    order with buffers of the same shape:

      1 all-gather      packed session rows [cap, ld_head]                    -> [W * cap, ld_head]
      2 all-gather      softmax statistics of the shard [W * cap, 3]          -> [W, W * cap, 3]
      3 reduce-scatter  dX partial [W * cap, ek]  (all-reduce + slice where the backend has no reduce-scatter)
      4 all-gather      packed item-row gradients [cap * T, ldr]              -> [W * cap * T, ldr]
      5 all-reduce      arena gradients + norm pieces
      6 all-gather      updated item rows of the shard [S, ldh]               -> [W, S, ldh]   (update steps only)

    Collective 6 is issued asynchronously: nothing of the step that follows needs the other shards' rows before its session
    gathers, so `wait_rows()` is called at the top of the next step (and by every other reader of the item table)."""

    def __init__(self, group=None, world: Optional[int] = None, rank: Optional[int] = None, sim: bool = False,
                 reduce_scatter: Optional[bool] = None, force: Optional[bool] = None, direct: Optional[bool] = None):
        live = dist.is_available() and dist.is_initialized()
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if live else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if live else 0)
        self.sim = sim
        # force (TCAR_FORCE_COLLECTIVES=1): a process group of ONE rank still issues every collective of the schedule — the code
        # an N-rank job runs (RCCL calls, their stream, the staging copies), with identity results.  Default: a single rank
        # short-circuits them.
        if force is None:
            force = bool(int(os.environ.get("TCAR_FORCE_COLLECTIVES", "0") or 0))
        self.collective = live and not sim and (self.world > 1 or bool(force))
        self.backend = dist.get_backend(group) if self.collective else "none"
        # RCCL called directly on the step's OWN stream (rccl.py) instead of through the process group.  Not for the host time of a
        # call (measured: 10 us against ~30) but for the STREAMS: ProcessGroupNCCL runs every collective on a stream of its own, and
        # the step already keeps four busy (main, aux, third, the sampler's).  HIP multiplexes streams onto 4 hardware queues by
        # default; with a fifth, two of them share a queue and — which two depends on creation order — the step ran at 1.6-1.9 ms
        # instead of 0.65-0.73 (profiles/r06_ab_experiments.txt, GPU_MAX_HW_QUEUES = 4 / 6 / 8 / 12).  Direct calls add NO stream:
        # every collective, also the item-row all-gather of collective 6, is issued on the stream the step runs on.  `direct`: None
        # = default (on for the nccl backend, TCAR_RCCL_DIRECT=0 switches it off), False = process group, True = required.
        self.direct = None
        if self.collective and self.backend == "nccl" and direct is not False:
            from . import rccl
            self.direct = rccl.make_direct(group, n=1)
            if direct is True and self.direct is None:
                raise RuntimeError("the direct RCCL path was required and could not be built")
        self.use_reduce_scatter = (self.backend == "nccl") if reduce_scatter is None else bool(reduce_scatter)
        self.bytes_moved: Dict[str, int] = {}
        self.order = []                 # names of the collectives in issue order (tests)
        self._pending_rows = None
        # optional per-collective timing (bench.py's second, event-instrumented pass; never in the headline pass: an event pair
        # costs its stream a few us): key -> list of (start, end) — CUDA events on the issuing stream, or host seconds on CPU
        self.timing = False
        self._times: Dict[str, list] = {}

    def _timed(self, key, t, fn):
        """run collective `fn` and, when timing is on, bracket it (for an async collective: the issue only)"""
        if not self.timing:
            return fn()
        if t.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn()
            e1.record()
            self._times.setdefault(key, []).append((e0, e1))
        else:
            import time
            t0 = time.perf_counter()
            out = fn()
            self._times.setdefault(key, []).append((t0, time.perf_counter()))
        return out

    def collective_ms(self) -> Dict[str, float]:
        """mean milliseconds per collective over the timed steps (call after a device synchronise)"""
        out = {}
        for key, spans in self._times.items():
            if not spans:
                continue
            if isinstance(spans[0][0], float):
                out[key] = round(1e3 * sum(b - a for a, b in spans) / len(spans), 4)
            else:
                out[key] = round(sum(a.elapsed_time(b) for a, b in spans) / len(spans), 4)
        return out

    def _note(self, key, t):
        self.bytes_moved[key] = t.numel() * t.element_size()
        self.order.append(key)

    def allgather(self, t: torch.Tensor, key: str) -> torch.Tensor:
        """[..] -> [W, ..] (rank-major); world 1: a view"""
        if self.sim:
            return t.unsqueeze(0).expand((self.world,) + tuple(t.shape)).contiguous()
        if not self.collective:
            return t.unsqueeze(0)
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        if self.direct is not None:
            self._timed(key, t, lambda: self.direct[0].all_gather(t.contiguous(), out))
        else:
            self._timed(key, t, lambda: dist.all_gather_into_tensor(out.view(-1), t.reshape(-1).contiguous(), group=self.group))
        self._note(key, out)
        return out

    def reduce_scatter_rows(self, full: torch.Tensor, cap: int, key: str) -> torch.Tensor:
        """sum over the ranks of full [W*cap, C]; returns this rank's rows [cap, C]"""
        if not self.collective:
            return full[:cap]
        self._note(key, full)
        if self.direct is not None:
            out = torch.empty(cap, full.shape[1], dtype=full.dtype, device=full.device)
            self._timed(key, full, lambda: self.direct[0].reduce_scatter(full.contiguous(), out))
            return out
        if self.use_reduce_scatter:
            out = torch.empty(cap, full.shape[1], dtype=full.dtype, device=full.device)
            self._timed(key, full, lambda: dist.reduce_scatter_tensor(out, full, group=self.group))
            return out
        self._timed(key, full, lambda: dist.all_reduce(full, group=self.group))     # gloo (CPU tests, single-GPU dry runs) has no reduce-scatter
        return full[self.rank * cap:(self.rank + 1) * cap]

    def allreduce(self, t: torch.Tensor, key: str) -> torch.Tensor:
        if self.collective:
            if self.direct is not None:
                self._timed(key, t, lambda: self.direct[0].all_reduce(t))
            else:
                self._timed(key, t, lambda: dist.all_reduce(t, group=self.group))
            self._note(key, t)
        return t

    def share_rows(self, stage: torch.Tensor, install) -> None:
        """collective 6: `stage` [W, S, ldh] holds this rank's updated rows in slot `rank`; `install(stage)` copies the gathered
        table into place once the collective has landed (wait_rows)"""
        if not self.collective:
            return
        if self.direct is not None:
            # on the step's own stream: nothing of this rank runs between the update and the next step's gathers, which need the
            # rows anyway — a side stream would buy no overlap and cost a hardware queue (see __init__).  IN PLACE: this rank's rows
            # already sit in their slot of the gathered table (RCCL: sendbuff == recvbuff + rank * sendcount), no staging copy
            mine = stage[self.rank].reshape(-1)
            self._timed("item_rows (issue)", stage, lambda: self.direct[0].all_gather(mine, stage.view(-1)))
            work = None
        else:
            mine = stage[self.rank].reshape(-1).clone()
            work = self._timed("item_rows (issue)", stage,
                               lambda: dist.all_gather_into_tensor(stage.view(-1), mine, group=self.group, async_op=True))
        self._note("item_rows", stage)
        self._pending_rows = (work, stage, install, mine)

    def wait_rows(self) -> None:
        if self._pending_rows is not None:
            work, stage, install, _ = self._pending_rows
            self._pending_rows = None
            if work is not None:
                work.wait()                   # NCCL: orders the current stream behind the collective, no host block
            install(stage)

    def step(self, pieces, cap: int, update: bool) -> None:
        """one training step; `pieces` supplies the local compute between the exchanges:
             begin() -> head [cap, ld]             score(head_all [W*cap, ld]) -> stats [W*cap, 3]
             backward(stats_all [W, W*cap, 3]) -> dx_full [W*cap, ek]         (dE of the shard stays inside)
             finish()                                                          (negative rows, shard norm, candidate-time grads)
             session_backward(dx_rows [cap, ek]) -> rows [cap*T, ldr]
             scatter(all_rows [W*cap*T, ldr])     arena() -> flat tensor      norms()
             update() -> (stage [W, S, ldh], install) or None"""
        self.order = []
        self.wait_rows()
        head = pieces.begin()
        head_all = self.allgather(head, "attout+labels+negatives").view(self.world * cap, -1)
        stats = pieces.score(head_all)
        stats_all = self.allgather(stats, "softmax_stats")
        dx_full = pieces.backward(stats_all)
        pieces.finish()
        dx_rows = self.reduce_scatter_rows(dx_full, cap, "dX")
        rows = pieces.session_backward(dx_rows)
        all_rows = self.allgather(rows, "rows+ids").view(-1, rows.shape[-1])
        pieces.scatter(all_rows)
        self.allreduce(pieces.arena(), "arena")
        pieces.norms()
        if update:
            out = pieces.update()
            if out is not None:
                self.share_rows(*out)


class _Pieces:
    """The local compute between the exchanges of ShardExchange.step, bound to the C entry points of csrc/step.hip.  ONE object per
    engine, re-armed per step (a class statement and eight closures per step were host time the step cannot hide behind)."""

    def __init__(self, eng):
        self.eng = eng

    def arm(self, bt, cap, K, nr, ldr, refresh, lr_pending, ctx, sctx, sh):
        e = self.eng
        self.bt, self.cap, self.K, self.nr, self.ldr, self.refresh, self.lr_pending = bt, cap, K, nr, ldr, refresh, lr_pending
        self.ctx, self.sctx, self.sh, self.st = ctx, sctx, sh, e._stream()
        self.B = bt.B if bt is not None else 0
        self.Bq = e.world * cap
        self.has_neg = K > 0

    # zero the arena, session forward, ONE packed row per session: [attout (ek) | label | coefficient of the negative
    # term | K negatives | pad], ints as bits, row stride ld_head (tcar_shard_begin packs, tcar_shard_score unpacks)
    def begin(self):
        e, bt = self.eng, self.bt
        head = e.head_loc[:self.cap]
        check(e.lib.tcar_shard_begin(C.byref(self.ctx), C.byref(bt) if bt is not None else None, self.cap, e._kcap, e._p(head),
                                     e.ld_head, self.refresh, e.nl, self.lr_pending, self.st), "tcar_shard_begin")
        return head

    # scoring of the shard against every session + softmax statistics of the shard
    def score(self, head_all):
        e, sh = self.eng, self.sh
        sh.att_all, sh.ld_att, sh.head_K = head_all.data_ptr(), e.ld_head, (self.K if self.has_neg else 0)
        self._head_all = head_all
        tk = e._tick3(0)
        check(e.lib.tcar_shard_score(C.byref(self.sctx), C.byref(sh), self.refresh, self.st), "tcar_shard_score")
        e._tock3(tk)
        if self.refresh:        # (the one-hot schedule issues no refresh: the planes stay dirty for the next op-level reader)
            e._time_dirty = False
        return e.s_stats[:self.Bq]

    # lse, dlogits planes, dE of the shard (aux stream, stays here), dX partial (goes home)
    def backward(self, stats_all):
        e = self.eng
        self._stats_all = stats_all
        tk = e._tick3(1)
        check(e.lib.tcar_shard_backward(C.byref(self.sctx), C.byref(self.sh), e._p(stats_all), self.st), "tcar_shard_backward")
        e._tock3(tk)
        return e.s_dx[:self.Bq]

    # negative rows, shard norm, candidate-time backward: on the aux stream behind dE, beside the dX exchange and the
    # session backward (tcar_shard_join orders the main stream behind them)
    def finish(self):
        e = self.eng
        check(e.lib.tcar_shard_finish(C.byref(self.sctx), C.byref(self.sh), self.K if self.has_neg else 0,
                                      e._p(e.s_neg) if self.has_neg else None, e._p(e.s_coef) if self.has_neg else None, self.st),
              "tcar_shard_finish")

    # session backward (local) -> packed rows [row (ldh) | id | pad]
    def session_backward(self, dx_rows):
        e, bt = self.eng, self.bt
        rows = e._rows_buf[:self.nr]
        if bt is not None:
            if not dx_rows.is_contiguous():
                dx_rows = dx_rows.contiguous()
            ce_rows = e.s_ce[e.dp_rank * self.cap:e.dp_rank * self.cap + self.B]
            check(e.lib.tcar_step_session_backward(C.byref(self.ctx), C.byref(bt), e._p(dx_rows), e._p(rows), self.ldr, self.nr,
                                                   e._p(ce_rows), self.st), "tcar_step_session_backward")
        else:
            rows.zero_()                    # an empty rank contributes padding rows only (id 0, zero payload)
        return rows

    # ids are 1-based: the rows of this shard become 1 .. nl, the rest (and the id-0 padding) fall out
    def scatter(self, all_rows):
        e = self.eng
        check(e.lib.tcar_shard_join(C.byref(self.ctx), self.st), "tcar_shard_join")
        check(e.lib.tcar_scatter_add_rows_packed(C.byref(e.dims_cand), e._p(all_rows), self.ldr, all_rows.shape[0], e.n0, e._p(e.Gi),
                                                 self.st), "tcar_scatter_add_rows_packed")

    # arena gradients + norm pieces incl. the shards' dense item norms
    def arena(self):
        return self.eng.Gx

    # dense-weight norms, summed in a fixed order (one workgroup per variable): identical gradients give identical
    # norms on every rank, the replicas stay bit-identical without a broadcast
    def norms(self):
        check(self.eng.lib.tcar_step_dense_norms(C.byref(self.ctx), self.st), "tcar_step_dense_norms")

    def update(self):
        self.eng.poll_fork_errors()              # a fork of this step that has already timed out: raise BEFORE the variables move
        return self.eng._update_local()


class ShardedEngine(TcarEngine):
    def __init__(self, params, content_emb, mwdhm, lr=1e-3, max_grad=150.0, neg_weight=0.01, device="cuda:0", group=None,
                 scoring="bf16x3", world: Optional[int] = None, rank: Optional[int] = None,
                 force_collectives: Optional[bool] = None, direct_rccl: Optional[bool] = None, **kw):
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
        if force_collectives is None:
            force_collectives = bool(int(os.environ.get("TCAR_FORCE_COLLECTIVES", "0") or 0))
        if live and not self._sim and (self.world > 1 or force_collectives):
            self.priority_stream = False          # (TcarEngine.priority_stream: no priority stream beside live collectives)
        super().__init__(params, content_emb, mwdhm, lr=lr, max_grad=max_grad, neg_weight=neg_weight, device=device,
                         scoring=scoring, shard=(n0, nl), **kw)
        self.n0, self.nl = n0, nl
        self.nlpad = _ru(nl, 128)
        self.n_local_items = nl
        # the collective schedule (device-agnostic; dX is reduce-scattered where the backend can — RCCL — and all-reduced over gloo)
        self.xch = ShardExchange(group, self.world, self.dp_rank, sim=self._sim, force=force_collectives, direct=direct_rccl)
        self.backend = self.xch.backend
        if self.xch.collective and self.tune is None and not os.environ.get("TCAR_SHARD_ALL_FLAGS"):
            # Flag forks whose PRODUCER sits behind a collective: the gather behind the item-row all-gather of the
            # previous update (FK_GATHER), the slab reduce and the input-gradient launch behind the dX reduce-scatter (FK_REDUCE,
            # FK_INGRAD).  A peer whose host stalls holds those producers back for as long as it likes, and a poll gives up after
            # 1 s — so these three fork through events whenever collectives are live; the others depend on local work only.
            mask = int(_lib.tuning().flag_fork) & ~((1 << 3) | (1 << 4) | (1 << 6))
            self.set_tuning(TCAR_FLAG_FORK=mask)
        self.cap = 0
        self._stage = torch.zeros(self.world, self.S, self.geo.ldh, dtype=torch.float32, device=self.dev)
        self.bytes_moved = self.xch.bytes_moved

    # ------------------------------------------------------------------------------------------ collectives
    def exchange_info(self) -> Dict[str, object]:
        g = self.geo
        return {"mode": "sharded", "world": self.world, "shard_rows": self.S,
                "collectives": ("none" if not self.xch.collective else
                                "RCCL C API on the step's stream (rccl.py)" if self.xch.direct is not None else
                                "torch.distributed process group (%s)" % self.xch.backend),
                "bytes_per_step": dict(self.bytes_moved),
                "ms_per_collective": self.xch.collective_ms(),      # filled by the event-instrumented pass (enable_native_timing)
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
            # the benchmarked single-GPU schedule on the shard (tcar_hip.h: tcar_shard_score): softmax epilogue + one-hot forms —
            # no fp32 logits; their workspaces are sized for the W * cap session rows of the exchange
            self.onehot = (self.scoring_code == 3 and self.scoring_bwd == 1 and g.ldt == 64
                           and not os.environ.get("TCAR_SHARD_MATERIALISED") and not os.environ.get("TCAR_NO_ONEHOT")
                           and not os.environ.get("TCAR_NO_ONEHOT_BWD"))
            if self.onehot:
                self.s_logits = None
                need = Bq * ((self.nl + 63) // 64 + 8) * 2 + 4 * Bq + 8
                self.s_ce_ws = torch.empty(need, **f32)
                self.s_ce_geo = (C.c_int32 * 2)(0, 0)
                if getattr(self, "s_oh16", None) is None:
                    self.s_oh16 = torch.empty(self.nlpad * 160, dtype=torch.bfloat16, device=self.dev)
                    check(self.lib.tcar_time_onehot(C.byref(self.dims_cand), self._p(self.mwdhm), self._p(self.s_oh16), 160,
                                                    self._stream()), "tcar_time_onehot")
                    self.s_tclip = torch.zeros(160 * g.ldt + 320, **f32)
                    self.s_qz = torch.zeros(5 * self.nl * 2, **f32)
                self.s_p16h, self.s_p16l = torch.zeros(Bp * 160, **bf), torch.zeros(Bp * 160, **bf)
                self.s_dP = torch.zeros(Bq * 160, **f32)
            else:
                self.s_logits = torch.empty(Bq, self.nlpad, **f32)
            self._sctx_src = None            # the shard context carries these pointers
            self.s_stats = torch.empty(Bq, 3, **f32)
            self.s_lse, self.s_ce = torch.empty(Bq, **f32), torch.empty(Bq, **f32)
            self.s_a16h, self.s_a16l = torch.zeros(Bp, g.ek, **bf), torch.zeros(Bp, g.ek, **bf)
            self.s_ap16h, self.s_ap16l = torch.zeros(Bp, g.ldh + g.pt, **bf), torch.zeros(Bp, g.ldh + g.pt, **bf)
            self.s_dl16h, self.s_dl16l = torch.zeros(Bp, self.nlpad, **bf), torch.zeros(Bp, self.nlpad, **bf)
            self.s_slabs = torch.empty(self.splitk, Bq, g.ek, **f32)
            self.s_dx = torch.empty(Bq, g.ek, **f32)
            # anchored softmax form on the shard (tcar_shard_t.aps16h / scale2): the per-row scaled attout plane of dE, the row scales
            self.s_aps16h = torch.zeros(Bp, g.ldh + g.pt, **bf) if getattr(self, "onehot", False) and not os.environ.get("TCAR_NO_CE_ANCHOR") else None
            self.s_scale2 = torch.zeros(2 * Bp, **f32) if self.s_aps16h is not None else None
            self.cap, self._kcap = cap, kc

    # two composite spans instead of the three GEMMs of the fused step: the C++ pieces between the exchanges
    TIMED_KERNELS = ("shard_score", "shard_backward")

    # ------------------------------------------------------------- timing of the scoring pieces (bench.py)
    def enable_native_timing(self, n: int):
        self._tm = {"n": n, "ev": [[], [], []]}
        self.xch.timing = True

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
    def _step(self, bt: Optional[Batch], cap: int, K: int, update: bool, T: int, lr_pending: float = -1.0):
        """one training step; bt = None: a rank whose shard of the global batch is empty still joins every collective (T is
        the step's input length: the row buffers of the exchanges have the same shape on every rank).  The collectives are
        ShardExchange.step; the pieces between them are the C entry points of csrc/step.hip (`_Pieces`).  lr_pending >= 0 (one
        rank): the previous step's update is owed and rides in tcar_shard_begin as the split update of the single-GPU step.
        A rank that exchanges nothing (one rank, no live collectives) makes ONE call: tcar_shard_step_local sequences the same
        pieces in C++ (round 6: the Python sequencing was ~0.2 ms of host time per step)."""
        g = self.geo
        B = bt.B if bt is not None else 0
        cap = max(cap, B, 1)
        self.xch.wait_rows()                       # the previous update's rows, before anything reads the item table
        self._ensure_work(max(B, 1), T)
        self._ensure_score(cap, K)
        nr = cap * T
        ldr = g.ldh + 4
        if getattr(self, "_rows_cap", 0) < nr:
            self._rows_buf = torch.zeros(nr, ldr, dtype=torch.float32, device=self.dev)
            self._rows_cap = nr
        # (the one-hot schedule reads the shard's time planes nowhere; eval_step rebuilds the fp32 time columns itself)
        refresh = 0 if self.onehot else int(self._time_dirty)
        ctx, sctx = self._ctx(), self._shard_ctx()
        sh = self._shard_desc(cap)
        if (bt is not None and not self.xch.collective and not self._sim and self.world == 1 and getattr(self, "_tm", None) is None
                and not os.environ.get("TCAR_SHARD_PY_STEP")):
            sh.att_all, sh.ld_att, sh.head_K = self.head_loc.data_ptr(), self.ld_head, K
            lr_u = -1.0
            if update:
                self.poll_fork_errors()
                lr_u = self._lr_t()
            check(self.lib.tcar_shard_step_local(C.byref(ctx), C.byref(sctx), C.byref(sh), C.byref(bt), self._kcap,
                                                 C.c_void_p(self.head_loc.data_ptr()), self.ld_head, refresh, lr_pending,
                                                 C.c_void_p(self._rows_buf.data_ptr()), ldr, nr, C.byref(self.dims_cand), lr_u,
                                                 self._stream()), "tcar_shard_step_local")
            if refresh:
                self._time_dirty = False
            if update:
                self._after_update()
            return
        pc = getattr(self, "_pieces", None)
        if pc is None:
            pc = self._pieces = _Pieces(self)
        pc.arm(bt, cap, K, nr, ldr, refresh, lr_pending, ctx, sctx, sh)
        self.xch.step(pc, cap, update)

    def _shard_desc(self, cap: int) -> "_lib.Shard":
        key = (cap, self.s_dl16h.data_ptr())
        if getattr(self, "_sh_key", None) != key:
            sh = _lib.Shard()
            sh.world, sh.cap, sh.n0, sh.n_loc = self.world, cap, self.n0, self.nl
            if self.s_logits is not None:
                sh.logits = self.s_logits.data_ptr()
            for n, t in (("stats", self.s_stats), ("lse", self.s_lse), ("ce", self.s_ce),
                         ("a16h", self.s_a16h), ("a16l", self.s_a16l), ("ap16h", self.s_ap16h), ("ap16l", self.s_ap16l),
                         ("dl16h", self.s_dl16h), ("dl16l", self.s_dl16l), ("slabs", self.s_slabs), ("dx", self.s_dx),
                         ("lab_all", self.s_lab), ("neg_all", self.s_neg), ("coef_all", self.s_coef)):
                setattr(sh, n, t.data_ptr())
            if getattr(self, "s_aps16h", None) is not None:
                sh.aps16h, sh.scale2, sh.n_total = self.s_aps16h.data_ptr(), self.s_scale2.data_ptr(), self.geo.N
            self._sh, self._sh_key = sh, key
        return self._sh

    def shard_form(self, cap: int) -> Dict[str, bool]:
        """the form the C++ shard pieces take at a capacity of `cap` sessions per rank (tcar_shard_form: their own predicates)"""
        self._ensure_score(cap, getattr(self, "_kcap", 0) or 0)
        out = (C.c_int32 * 2)()
        check(self.lib.tcar_shard_form(C.byref(self._shard_ctx()), C.byref(self._shard_desc(cap)), out), "tcar_shard_form")
        return {"onehot": bool(out[0]), "ce_anchored": bool(out[1])}

    def _shard_ctx(self):
        """tcar_ctx_t whose candidate side is this rank's shard (tcar_step_update: arena + the owned item rows + planes)"""
        c = self._ctx()
        if getattr(self, "_sctx_src", None) is not c:
            s = _lib.Ctx()
            C.memmove(C.byref(s), C.byref(c), C.sizeof(_lib.Ctx))
            s.d = self.dims_cand
            s.E = self.E.data_ptr() + 4 * self.n0 * self.geo.ek
            if getattr(self, "onehot", False):
                s.ce_ws, s.ce_ws_floats = self.s_ce_ws.data_ptr(), self.s_ce_ws.numel()
                s.ce_geo = C.cast(self.s_ce_geo, C.c_void_p)
                s.oh16, s.p16h, s.p16l = self.s_oh16.data_ptr(), self.s_p16h.data_ptr(), self.s_p16l.data_ptr()
                s.tclip, s.dP, s.qz = self.s_tclip.data_ptr(), self.s_dP.data_ptr(), self.s_qz.data_ptr()
            else:
                s.oh16 = s.tclip = s.dP = s.qz = None
            self._sctx, self._sctx_src = s, c
        return self._sctx

    def _update_local(self):
        """clip + Adam: arena on every rank, the item rows of the shard by their owner; returns the staging buffer of collective 6
        (this rank's updated rows in its slot) and the installer that copies the gathered table into E"""
        g = self.geo
        check(self.lib.tcar_step_update(C.byref(self._shard_ctx()), self._lr_t(), self._stream()), "tcar_step_update")
        self._after_update()
        if not self.xch.collective:
            return None
        self._stage[self.dp_rank, :self.nl].copy_(self.E[self.n0:self.n0 + self.nl, :g.ldh])

        def install(stage):
            self.E[:g.N, :g.ldh].copy_(stage.view(-1, g.ldh)[:g.N])
        return self._stage, install

    def _update_and_share(self):
        out = self._update_local()
        if out is not None:
            self.xch.share_rows(*out)

    def flush(self):
        """a deferred update (one rank, train_step(defer_update=True)) applied now + the pending item-row all-gather of the last
        update (every reader of the item table calls flush)"""
        if self._pending_lr is not None:
            lr, self._pending_lr = self._pending_lr, None
            check(self.lib.tcar_step_update(C.byref(self._shard_ctx()), lr, self._stream()), "tcar_step_update")
        if getattr(self, "xch", None) is not None:
            self.xch.wait_rows()

    @property
    def can_defer(self) -> bool:
        """the split (deferred) update of the single-GPU step applies when this rank owns the whole table and nothing is exchanged
        behind the update: ONE rank.  With more ranks the owned rows must reach the other ranks before their next gather, so the
        update (1 / world of the table) stays inside its step."""
        return self.world == 1 and not self._sim and not self.xch.collective and self.overlap and hasattr(self, "adam_bitmap")

    # ------------------------------------------------------------------------------------------- public API
    score_batch = property(lambda self: self.world * max(self.cap, 1))

    def train_step(self, batch, bt: Optional[Batch] = None, cap_rows: Optional[int] = None, cap: Optional[int] = None,
                   T: Optional[int] = None, K: Optional[int] = None, defer_update: bool = False):
        """`batch` / `bt` = this rank's sessions (None: its shard of the global batch is empty — it still joins every
        collective and must be told the step's input length T and negative count K, which every rank knows from the bucket
        schedule).  cap = rows every rank contributes to the all-gathers (>= the largest local batch of the step; default: the
        local batch size — weak-scaling runs with equal batches); cap_rows (dp.DPEngine's argument: cap * T) is accepted
        for drop-in use.  No metadata collective, no host synchronisation."""
        # defer_update (TcarEngine's split update): on ONE rank the update of this step is applied at the start of the next one
        # (tcar_shard_begin) or by flush(), as in the single-GPU step; with more ranks it is ignored — the sharded update is followed
        # by the all-gather of the updated rows, which the next step's gathers need: there is nothing to defer it behind
        if bt is None and batch is not None:
            bt = self.upload(batch)
        if bt is not None:
            T = bt.T
            K = bt.K if bt.neg else 0
        elif T is None or K is None:
            raise ValueError("an empty rank needs the step's T and K")
        if cap is None:
            cap = (cap_rows // max(T, 1)) if cap_rows else (bt.B if bt is not None else 1)
        if bt is not None and self.can_defer and (defer_update or self._pending_lr is not None):
            lr_p, self._pending_lr = (self._pending_lr if self._pending_lr is not None else -1.0), None
            self._step(bt, cap, K, False, T, lr_pending=lr_p)
            self._pending_lr = self._lr_t()
            self._after_update()                  # step count / beta powers advance now; the device work is owed
            self.poll_fork_errors()
            if not defer_update:
                self.flush()
            return self._loss_rows(bt, cap, K)
        self.flush()
        self._step(bt, cap, K, True, T)
        self.poll_fork_errors()                   # (a flag fork of this step's pieces that timed out: never silent)
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
        if self.xch.collective:
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
