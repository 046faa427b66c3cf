"""RCCL's C API through ctypes: the collectives of the two exchanges issued DIRECTLY on the stream the step runs on.

Why (round 6, `profiles/r06_ab_experiments.txt` section 4): `torch.distributed` runs every collective on a stream of
ProcessGroupNCCL's own.  The training step already keeps four streams busy (main, aux, third, the sampler's) and HIP multiplexes
streams onto FOUR hardware queues by default: with a fifth stream two of them share a queue, and — which two depends on creation
order and on GPU_MAX_HW_QUEUES — the catalog-sharded step with live collectives ran at 1.6-1.9 ms instead of 0.65-0.73 ms on a
world-1 communicator.  `ncclAllGather` / `ncclReduceScatter` / `ncclAllReduce` issued on OUR stream add no stream: they are
stream-ordered behind the kernels that produced their input and in front of the kernels that read their output, no event, no hop
(and one C call each: ~10 us of host time against ~30 through the process group).

The communicator is built once per exchange from a `ncclUniqueId` that rank 0 creates and `torch.distributed` broadcasts (the
process group stays the rendezvous and the fallback: `selftest()` runs the three collectives on tiny tensors against their known
answers on every rank, and the exchanges fall back to the process group — on ALL ranks, agreed through the group — if the direct
path cannot be built or answers wrongly).  `librccl.so` is the copy bundled with PyTorch-ROCm (the one the `nccl` backend itself
uses), so both paths share one RCCL.

Not a compatibility layer: RCCL is the only backend this talks to."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
import torch.distributed as dist

NCCL_UNIQUE_ID_BYTES = 128
_DT = {torch.float32: 7, torch.int32: 2, torch.uint8: 1, torch.int64: 4, torch.bfloat16: 9, torch.float64: 8}
_SUM, _MAX = 0, 2


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


class RcclError(RuntimeError):
    pass


_LIB = None


def _lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = C.CDLL(path if os.path.exists(path) else "librccl.so")
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(vp), i32, _UniqueId, i32]
        lib.ncclCommDestroy.argtypes = [vp]
        lib.ncclAllGather.argtypes = [vp, vp, sz, i32, vp, vp]
        lib.ncclReduceScatter.argtypes = [vp, vp, sz, i32, i32, vp, vp]
        lib.ncclAllReduce.argtypes = [vp, vp, sz, i32, i32, vp, vp]
        lib.ncclGetErrorString.argtypes = [i32]
        lib.ncclGetErrorString.restype = C.c_char_p
        for f in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclReduceScatter", "ncclAllReduce"):
            getattr(lib, f).restype = i32
        _LIB = lib
    return _LIB


def _ck(rc: int, what: str):
    if rc != 0:
        raise RcclError("%s failed: %s" % (what, _lib().ncclGetErrorString(rc).decode(errors="replace")))


class RcclComm:
    """One RCCL communicator over the ranks of `group` (a COLLECTIVE constructor: every rank of the group calls it, on the device
    it computes on).  Tensors must be contiguous and live on that device; `stream` defaults to torch's current stream."""

    def __init__(self, group=None, device=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RcclError("no process group to exchange the communicator id through")
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        lib = _lib()
        uid = _UniqueId()
        if self.rank == 0:
            _ck(lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        on_dev = dist.get_backend(group) == "nccl"
        raw0 = C.string_at(C.addressof(uid), NCCL_UNIQUE_ID_BYTES) if self.rank == 0 else bytes(NCCL_UNIQUE_ID_BYTES)
        buf = torch.tensor(list(raw0), dtype=torch.uint8)
        if on_dev:
            buf = buf.to(self.dev)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(buf, src, group=group)
        raw = bytes(buf.cpu().tolist())
        C.memmove(C.addressof(uid), raw, NCCL_UNIQUE_ID_BYTES)
        self.comm = C.c_void_p()
        with torch.cuda.device(self.dev):
            _ck(lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")
        self._lib = lib

    def _st(self, stream):
        return C.c_void_p(stream if stream is not None else torch.cuda.current_stream(self.dev).cuda_stream)

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor, stream: Optional[int] = None):
        """recv [world * send.numel()] <- every rank's send (rank-major)"""
        assert send.is_contiguous() and recv.is_contiguous() and recv.numel() == self.world * send.numel() and send.dtype == recv.dtype
        _ck(self._lib.ncclAllGather(C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), send.numel(), _DT[send.dtype], self.comm,
                                    self._st(stream)), "ncclAllGather")

    def reduce_scatter(self, send: torch.Tensor, recv: torch.Tensor, stream: Optional[int] = None):
        """recv [send.numel() / world] <- this rank's block of the element-wise sum of every rank's send"""
        assert send.is_contiguous() and recv.is_contiguous() and send.numel() == self.world * recv.numel() and send.dtype == recv.dtype
        _ck(self._lib.ncclReduceScatter(C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), recv.numel(), _DT[send.dtype], _SUM,
                                        self.comm, self._st(stream)), "ncclReduceScatter")

    def all_reduce(self, t: torch.Tensor, op: int = _SUM, stream: Optional[int] = None):
        """in place"""
        assert t.is_contiguous()
        _ck(self._lib.ncclAllReduce(C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), t.numel(), _DT[t.dtype], op, self.comm,
                                    self._st(stream)), "ncclAllReduce")

    def selftest(self) -> bool:
        """the three collectives on tiny tensors against their known answers (as dp.preflight does through the process group)"""
        w, r, dev = self.world, self.rank, self.dev
        t = torch.full((4,), float(r + 1), device=dev)
        self.all_reduce(t)
        mine = torch.full((3,), float(r), device=dev)
        out = torch.empty(w * 3, device=dev)
        self.all_gather(mine, out)
        full = torch.arange(w * 2, dtype=torch.float32, device=dev) + r
        part = torch.empty(2, device=dev)
        self.reduce_scatter(full, part)
        torch.cuda.synchronize(dev)
        ok = bool((t == w * (w + 1) / 2.0).all()) and out.view(w, 3)[:, 0].tolist() == [float(i) for i in range(w)] and \
            part.tolist() == [w * (2 * r + j) + w * (w - 1) / 2.0 for j in range(2)]
        return ok

    def destroy(self):
        if getattr(self, "comm", None) is not None and self.comm.value:
            self._lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


def make_direct(group=None, device=None, n: int = 1, verbose: bool = True):
    """`n` direct communicators for the ranks of `group`, or None when the direct path is switched off (TCAR_RCCL_DIRECT=0), the
    backend is not RCCL, or any rank fails to build / verify it — the decision is all-reduced through the process group so that
    every rank takes the same path."""
    if os.environ.get("TCAR_RCCL_DIRECT", "1") == "0":
        return None
    if not (dist.is_available() and dist.is_initialized()) or dist.get_backend(group) != "nccl":
        return None
    comms, err = [], None
    try:
        for _ in range(n):
            comms.append(RcclComm(group, device))
        good = all(c.selftest() for c in comms)
    except Exception as e:                    # (construction is collective: a rank that fails here has already left the others
        good, err = False, e                  #  inside ncclCommInitRank — they time out there; nothing this function can mend)
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    flag = torch.tensor([1 if good else 0], device=dev, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) != 1:
        if verbose:
            import sys
            print("[tcar] direct RCCL path unavailable (%s): collectives through torch.distributed" % (err or "self-test mismatch"),
                  file=sys.stderr)
        for c in comms:
            try:
                c.destroy()
            except Exception:
                pass
        return None
    return comms
