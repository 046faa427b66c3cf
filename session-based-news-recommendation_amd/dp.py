"""Data-parallel TCAR step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference is single-process (no collectives anywhere); this layer is new design.  Examples are
independent given the weights and the loss is a SUM over sessions (model_combine.py:147,156), so the global
gradient is the plain sum of the ranks' gradients — no 1/W rescale — and every rank applies the same clip + Adam
to its replica.  What makes it more than one all-reduce is the clip: tf.clip_by_norm of an IndexedSlices uses the
norm of the concatenated slice VALUES, which is not additive for blocks that are sums over the
whole (global) batch.  The exchange has these steps (`GradExchange`; the collectives 1, 5, 3 are issued in that one
canonical order on every rank and do not depend on each other, the local steps 2, 4, 6 follow once they have landed):

  1. all-reduce  [ dE_item | dE_time ]      [N, ldh + pt] fp32 — the dense item-table block (scoring + densified
                                            negative part) and the candidate-side time block, BEFORE any per-row work
  2. local       ||dE_item||^2              -> dense norm piece of item_emb (now global)
  3. all-reduce  [ arena grads | pieces ]   dense weights, small tables (session-side rows only), per-row norm
                                            pieces (these ARE additive: each gathered row belongs to one rank)
  4. local       candidate-side clip backward of the reduced dE_time -> time-table grads + their norm pieces,
                                            added ONCE after step 3 (adding before would count them W times)
  5. all-gather  (item id, gradient row)    the "bucketed sparse-embedding exchange": B_local*T rows per rank,
                                            padded to a common row count with id 0 (skipped by the scatter)
  6. local       scatter-add all rows into dE_item, then the norms of the dense weights.

xGMI is point-to-point (7 links/GPU): step 1 moves ~106 MB per rank at the Globo size and dominates; it only
depends on dE, so the rank-local backward runs dE FIRST and `DPEngine` starts step 1 on a communication stream the
moment dE is complete (an event recorded by the C++ driver): the all-reduce runs beside dX, the attention / projection
backward and the weight gradients; the sparse-row all-gather and the arena all-reduce are queued behind it on the same
stream as soon as the rank-local backward has produced them, and the main stream joins once, before step 2.  Step 1 is
sent as two all-reduces, the candidate-time block first, so that step 4 can run (into a scratch block, added to the arena
after step 3) while the item block is still on the wire.
`GradExchange` is device-agnostic (tests run it over gloo on CPU tensors with the oracle's gradients).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import check
from .engine import SLOT, TcarEngine


class GradExchange:
    """The collective schedule above, independent of where the local pieces come from."""

    def __init__(self, group=None, force=None, direct=None):
        import os
        self.group = group
        live = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if live else 1
        # force (TCAR_FORCE_COLLECTIVES=1): a process group of ONE rank still issues every collective (identity results) — the
        # calls, streams and staging buffers of an N-rank job on the one GPU there is
        if force is None:
            force = bool(int(os.environ.get("TCAR_FORCE_COLLECTIVES", "0") or 0))
        self.collective = live and (self.world > 1 or bool(force))
        # RCCL directly on the issuing stream instead of through the process group (rccl.py; sharded.ShardExchange has the account)
        self.direct = None
        if self.collective and dist.get_backend(group) == "nccl" and direct is not False:
            from . import rccl
            got = rccl.make_direct(group, n=1)
            self.direct = got[0] if got else None
            if direct is True and self.direct is None:
                raise RuntimeError("the direct RCCL path was required and could not be built")

    def communicate(self, big: torch.Tensor, arena_pieces: torch.Tensor, ids: torch.Tensor, rows: torch.Tensor,
                    big_done: bool = False):
        """Every collective of the step, in ONE canonical order on every rank (also the ranks whose shard is empty):
        all-reduce big (1) -> all-gather ids, rows (5) -> all-reduce arena (3).  None of them depends on another one's
        result, so a caller may issue them back to back on a communication stream.  `big_done`: (1) was already issued."""
        g = self.group
        if not self.collective:
            return ids.reshape(-1), rows.reshape(-1, rows.shape[-1])
        d = self.direct
        if not big_done:
            for part in (big if isinstance(big, (list, tuple)) else (big,)):
                self.all_reduce(part)                                       # 1  (parts in the caller's canonical order)
        all_ids = torch.empty((self.world,) + tuple(ids.shape), dtype=ids.dtype, device=ids.device)
        all_rows = torch.empty((self.world,) + tuple(rows.shape), dtype=rows.dtype, device=rows.device)
        if d is not None:
            d.all_gather(ids.reshape(-1).contiguous(), all_ids.view(-1))    # 5
            d.all_gather(rows.reshape(-1).contiguous(), all_rows.view(-1))
        else:
            dist.all_gather_into_tensor(all_ids.view(-1), ids.reshape(-1).contiguous(), group=g)      # 5
            dist.all_gather_into_tensor(all_rows.view(-1), rows.reshape(-1).contiguous(), group=g)
        self.all_reduce(arena_pieces)                                       # 3
        return all_ids.view(-1), all_rows.view(-1, rows.shape[-1])

    def all_reduce(self, t: torch.Tensor):
        """in-place sum over the ranks on torch's CURRENT stream (direct RCCL) / through the process group"""
        if self.direct is not None:
            self.direct.all_reduce(t)                  # (asserts contiguity: an in-place collective on a copy would be lost)
        else:
            dist.all_reduce(t, group=self.group)

    @staticmethod
    def finish(all_ids: torch.Tensor, all_rows: torch.Tensor, sqnorm_item: Callable[[], None],
               cand_time_bwd: Callable[[], None], scatter_rows: Callable[[torch.Tensor, torch.Tensor], None],
               sqnorm_dense: Callable[[], None]):
        """The local steps, once every collective has landed."""
        sqnorm_item()                                                       # 2  (before any row is scattered in: S5)
        cand_time_bwd()                                                     # 4  (once, on top of the reduced arena)
        scatter_rows(all_ids, all_rows)                                     # 6
        sqnorm_dense()

    def run(self, big: torch.Tensor, arena_pieces: torch.Tensor, ids: torch.Tensor, rows: torch.Tensor,
            sqnorm_item: Callable[[], None], cand_time_bwd: Callable[[], None],
            scatter_rows: Callable[[torch.Tensor, torch.Tensor], None], sqnorm_dense: Callable[[], None]):
        all_ids, all_rows = self.communicate(big, arena_pieces, ids, rows)
        self.finish(all_ids, all_rows, sqnorm_item, cand_time_bwd, scatter_rows, sqnorm_dense)


def preflight(group=None, device=None, verbose: bool = True) -> Dict[str, object]:
    """First contact with the process group: ONE tiny all_gather_into_tensor, reduce_scatter_tensor and all_reduce (the three
    collectives the two exchanges use) on `device`, checked against their known answers, before any large buffer exists.  A job
    whose collectives cannot run fails HERE with the backend, world size, device and library versions in the message instead
    of inside step 1.  Returns the capabilities ({"reduce_scatter": bool, ...}); `reduce_scatter` False (gloo has none) makes
    ShardedEngine reduce dX with an all-reduce."""
    import sys
    if not (dist.is_available() and dist.is_initialized()):
        return {"world": 1, "backend": "none", "reduce_scatter": False}
    world, rank, backend = dist.get_world_size(group), dist.get_rank(group), dist.get_backend(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    info = {"world": world, "rank": rank, "backend": backend, "device": str(dev), "torch": torch.__version__,
            "hip": getattr(torch.version, "hip", None), "reduce_scatter": False}
    try:
        if backend == "nccl":
            info["rccl"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as e:                                   # version query only: never fatal
        info["rccl"] = "unknown (%s)" % type(e).__name__
    step = "all_reduce"
    try:
        t = torch.full((4,), float(rank + 1), device=dev)
        dist.all_reduce(t, group=group)
        want = world * (world + 1) / 2.0
        if not bool((t == want).all()):
            raise RuntimeError("all_reduce returned %r, expected %r" % (t.tolist(), want))
        step = "all_gather_into_tensor"
        mine = torch.full((3,), float(rank), device=dev)
        out = torch.empty(world * 3, device=dev)
        dist.all_gather_into_tensor(out, mine, group=group)
        if out.view(world, 3)[:, 0].tolist() != [float(r) for r in range(world)]:
            raise RuntimeError("all_gather_into_tensor returned %r" % out.tolist())
        step = "reduce_scatter_tensor"
        try:
            full = torch.arange(world * 2, dtype=torch.float32, device=dev) + rank
            part = torch.empty(2, device=dev)
            dist.reduce_scatter_tensor(part, full, group=group)
            exp = [world * (2 * rank + j) + world * (world - 1) / 2.0 for j in range(2)]
            if part.tolist() != exp:
                raise RuntimeError("reduce_scatter_tensor returned %r, expected %r" % (part.tolist(), exp))
            info["reduce_scatter"] = True
        except (RuntimeError, NotImplementedError) as e:
            if backend == "nccl":
                raise
            info["reduce_scatter_error"] = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:120])
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
    except Exception as e:
        raise RuntimeError("collective preflight failed at %s (%s): %s: %s" % (step, info, type(e).__name__, e)) from e
    if verbose and rank == 0:
        print("[tcar] collective preflight ok: %s" % info, file=sys.stderr)
    return info


def shard_bounds(b: int, world: int, rank: int):
    """Contiguous split of a length-bucketed batch of b sessions (sampler.py:40-49) into `world` shards of
    ceil(b/world); trailing ranks may get fewer (or zero) rows.  Returns (lo, hi, cap)."""
    cap = (b + world - 1) // world
    lo = min(b, rank * cap)
    hi = min(b, lo + cap)
    return lo, hi, cap


class DPEngine(TcarEngine):
    """TcarEngine whose backward ends with the `GradExchange` schedule.  All ranks must call train_step with
    batches of the SAME input length T (they walk the same bucket schedule)."""

    flag_forks = False   # (the gradient exchange is enqueued inside the fused backward: event forks)

    def __init__(self, *a, group=None, force_collectives=None, direct_rccl=None, **kw):
        import os
        live = dist.is_available() and dist.is_initialized()
        fc = force_collectives if force_collectives is not None else bool(int(os.environ.get("TCAR_FORCE_COLLECTIVES", "0") or 0))
        if live and (dist.get_world_size(group) > 1 or fc):
            self.priority_stream = False          # (TcarEngine.priority_stream: no priority stream beside live collectives)
        super().__init__(*a, **kw)
        self.group = group
        self.xch = GradExchange(group, force=force_collectives, direct=direct_rccl)
        g = self.geo
        # step 1 in two parts, the candidate-time block FIRST: its clip backward can then run (into a scratch copy of the
        # time-table gradients) while the item block is still being reduced
        self.big_parts = [self.big[g.N * g.ldh:], self.big[:g.N * g.ldh]]
        self._ct_rows = 139 * g.ldt
        self._ct_scratch = torch.zeros(self._ct_rows + _lib.NSLOT, dtype=torch.float32, device=self.dev)
        self.rows_cap = 0

    def _ensure_rows(self, rows: int):
        if rows > self.rows_cap:
            self.rows_buf = torch.zeros(rows, self.geo.ldh, dtype=torch.float32, device=self.dev)
            self.ids_buf = torch.zeros(rows, dtype=torch.int32, device=self.dev)
            self.rows_cap = rows

    def finish_backward(self, bt, cap_rows: Optional[int] = None):
        g, lib, st, p = self.geo, self.lib, self._stream(), self._p
        B, T = (bt.B, bt.T) if bt is not None else (0, 0)
        n_rows = B * T
        cap = max(cap_rows or 0, n_rows)
        self._ensure_rows(max(cap, 1))
        rows, ids = self.rows_buf[:max(cap, 1)], self.ids_buf[:max(cap, 1)]
        if cap > n_rows:
            rows[n_rows:].zero_()
            ids[n_rows:].zero_()
        if bt is not None:
            ids[:n_rows].copy_(bt._seq_t[:n_rows])
            tab, gr = self._tables(), self._grads()
            gr.rows_out = rows.data_ptr()
            check(lib.tcar_gather_clip_bwd(C.byref(self.dims), C.byref(tab), C.byref(bt), p(self.dx_icp), p(self.dx_pt),
                                           p(self.dx_act), p(self.dclick), C.byref(gr), st), "tcar_gather_clip_bwd")

        def scatter(all_ids, all_rows):
            check(lib.tcar_scatter_add_rows(C.byref(self.dims), p(all_ids), p(all_rows), all_ids.numel(), p(self.Gi),
                                            self._stream()), "tcar_scatter_add_rows")

        if self._comm_busy:
            # the big all-reduce is already running on the communication stream (start_big_reduce); queue the other
            # collectives behind it there — they need only the arena / rows this stream has just produced — and join once
            main = torch.cuda.current_stream(self.dev)
            self._comm.wait_stream(main)
            with torch.cuda.stream(self._comm):
                all_ids, all_rows = self.xch.communicate(self.big_parts, self.Gx, ids, rows, big_done=True)
            main.wait_stream(self._comm)
            main.wait_stream(self._aux)                  # the candidate-time backward into the scratch block
            all_ids.record_stream(main)
            all_rows.record_stream(main)
            self._comm_busy = False
            # its table gradients and norm pieces go on top of the REDUCED arena (once, like step 4 of the in-line order)
            o = self.seg["month"]["off"]
            self.Gx[o:o + self._ct_rows] += self._ct_scratch[:self._ct_rows]
            self.sqn_pieces += self._ct_scratch[self._ct_rows:]
            self.xch.finish(all_ids, all_rows, self._sqnorm_item, lambda: None, scatter, self._sqnorm_dense)
            return
        all_ids, all_rows = self.xch.communicate(self.big_parts, self.Gx, ids, rows)
        aux = getattr(self, "_aux", None)
        if aux is not None and self.big.is_cuda:
            # the candidate-time backward (needs only the reduced d_et; adds atomically into the reduced arena) runs on the
            # aux stream beside the item norm, the row scatter and the dense-weight norms
            main = torch.cuda.current_stream(self.dev)
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                self._cand_time_bwd()
            self.xch.finish(all_ids, all_rows, self._sqnorm_item, lambda: None, scatter, self._sqnorm_dense)
            main.wait_stream(aux)
        else:
            self.xch.finish(all_ids, all_rows, self._sqnorm_item, self._cand_time_bwd, scatter, self._sqnorm_dense)

    _comm_busy = False
    async_exchanges = 0      # steps whose collectives ran on the communication stream (tests assert the path is live)

    def start_big_reduce(self):
        """Step 1 of the exchange, started on a communication stream as soon as dE is complete (event 3 of the C++
        driver, recorded after the dE GEMM + negative rows) so that it overlaps chain A.  Falls back to the in-line
        all-reduce when there is no aux stream / single rank."""
        self._comm_busy = False
        if not self.xch.collective or not self.big.is_cuda or not getattr(self, "_aux_ev", None):
            return
        if not hasattr(self, "_comm"):
            self._comm = torch.cuda.Stream(self.dev)
        self._comm.wait_event(self._aux_ev[3])
        with torch.cuda.stream(self._comm):
            self.xch.all_reduce(self.big_parts[0])                      # candidate-time block
            det_done = torch.cuda.Event()
            det_done.record(self._comm)
            self.xch.all_reduce(self.big_parts[1])                      # item block
        # candidate-side clip backward of the reduced time block, on the aux stream beside the item block's all-reduce,
        # into a scratch copy of the time-table gradients / norm pieces (added to the arena after ITS all-reduce)
        self._aux.wait_event(det_done)
        with torch.cuda.stream(self._aux):
            self._ct_scratch.zero_()
            gr = self._grads()
            base = self._ct_scratch.data_ptr()
            o0 = self.seg["month"]["off"]
            for k, n in enumerate(["month", "day", "week", "hour", "minute"]):
                gr.g_time[k] = base + 4 * (self.seg[n]["off"] - o0)
            gr.sqn = base + 4 * self._ct_rows
            check(self.lib.tcar_cand_time_bwd_indexed(C.byref(self.dims), C.byref(self._time_ptrs()), self._p(self.inv_n),
                                                      self._p(self.inv_off), self._p(self.d_et), int(self.scoring_code != 0),
                                                      self._p(self.ct_ws), C.byref(gr), self._stream()),
                  "tcar_cand_time_bwd_indexed")
        self._comm_busy = True
        self.async_exchanges += 1

    def train_step(self, batch, bt=None, cap_rows: Optional[int] = None, T: Optional[int] = None, K: Optional[int] = None):
        """`batch` may be None for a rank whose shard of the global batch is empty (it still joins the collectives).
        T / K: accepted for interface parity with sharded.ShardedEngine (an empty rank needs them there)."""
        if batch is None and bt is None:
            self.Gx.zero_()
            self.big.zero_()
            self.sqn_dense.zero_()
            self.finish_backward(None, cap_rows)
            self.update()
            return torch.zeros(0, device=self.dev)
        bt = bt or self.upload(batch)
        self._local(bt)
        self.finish_backward(bt, cap_rows)
        self.update()
        return self._loss_view(bt)

    def update(self):
        if self.native and self.timing is None:
            if self.work_B == 0:
                # a rank whose very first shard is empty has no workspace yet: give it a minimal one, so that the update
                # always goes through tcar_step_update (which also refreshes the bf16 planes of the item table)
                self._ensure_work(1, 1)
            check(self.lib.tcar_step_update(C.byref(self._ctx()), self._lr_t(), self._stream()), "tcar_step_update")
            self._after_update()
        else:
            super().update()

    def _local(self, bt):
        """forward + rank-local backward, driven from C++ (tcar_step_forward / tcar_step_backward_local)."""
        self.flush()
        if self.native and self.timing is None:
            self._ensure_work(bt.B, bt.T)
            ctx, st = self._ctx(), self._stream()
            check(self.lib.tcar_step_forward(C.byref(ctx), C.byref(bt), int(self._time_dirty), st), "tcar_step_forward")
            self._time_dirty = False
            check(self.lib.tcar_step_backward_local(C.byref(ctx), C.byref(bt), st), "tcar_step_backward_local")
            self.start_big_reduce()
        else:
            self.forward(bt)
            self.backward_local(bt)

    def loss_and_grads(self, batch, bt=None, cap_rows: Optional[int] = None):
        bt = bt or self.upload(batch)
        self._local(bt)
        self.finish_backward(bt, cap_rows)
        return self._loss_view(bt)


    def exchange_info(self) -> Dict[str, object]:
        """Bytes this rank hands to the collectives per step (bench.py prints it)."""
        g = self.geo
        big = 4 * g.N * (g.ldh + g.pt)
        arena = 4 * (self.arena_n + _lib.NSLOT)
        return {"mode": "replica", "world": self.xch.world, "allreduce_bytes": big + arena,
                "collectives": ("none" if not self.xch.collective else
                                "RCCL C API on the issuing stream (rccl.py)" if self.xch.direct is not None else
                                "torch.distributed process group"),
                "allgather_bytes_per_session_row": 4 * (g.ldh + 1),
                "note": "all-reduce of the dense item-table block + candidate-time block of dE, all-reduce of the arena, "
                        "all-gather of the (id, row) sparse item rows"}


def make_dp_engine(params, content_emb, mwdhm, device="cuda:0", group=None, scoring="bf16x3", mode="auto", **kw):
    """Data-parallel engine factory.  mode "replica": every rank holds the whole catalog and the dense item gradient is
    all-reduced (`DPEngine`); "sharded": catalog-sharded scoring (`sharded.ShardedEngine`, ~1/8 of the bytes); "auto"
    picks the sharded exchange when there is more than one rank."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if mode == "auto":
        if world > 1 and scoring != "f32":
            from . import sharded  # noqa: F401  (sets _HAVE_SHARDED)
        mode = "sharded" if (world > 1 and _HAVE_SHARDED and scoring != "f32") else "replica"
    if mode == "sharded":
        from .sharded import ShardedEngine
        return ShardedEngine(params, content_emb, mwdhm, device=device, group=group, scoring=scoring, **kw)
    return DPEngine(params, content_emb, mwdhm, device=device, group=group, scoring=scoring, **kw)


_HAVE_SHARDED = False        # flipped by sharded.py once the catalog-sharded engine is in place
