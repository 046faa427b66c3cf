"""HBM roofline of the embedding gather kernels at a size where launch latency does not dominate
(BASELINE.json north_star: >= 70 % of the HBM-read roofline on the embedding-gather kernel).
Usage: python tools/gather_bench.py [n_items] [B] [T]      (defaults: 2,000,000 items, B = 16384, T = 40)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tcar_amd  # noqa
from tcar_amd import _lib
from tcar_amd._lib import Batch, Dims, Grads, Tables

lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
T = int(sys.argv[3]) if len(sys.argv) > 3 else 40
H, Ht, ldh, ldt = 250, 64, 256, 64
ic, pt, ct, ek = 512, 320, 128, 832
dev = "cuda"
E = torch.randn(N, ek, device=dev) * 0.05
small = [torch.randn(v, ldt, device=dev) * 0.3 for v in (13, 32, 8, 25, 61, 11)]
pos = torch.randn(40, ldh, device=dev) * 0.02
g = torch.Generator(device="cpu").manual_seed(0)
ri = lambda lo, hi, *s: torch.randint(lo, hi, s, generator=g, dtype=torch.int32).to(dev)
seq = ri(1, N + 1, B, T)
pub = [ri(1, v, B, T) for v in (13, 32, 8, 25, 61)]
gap, cw, ch, label = ri(0, 11, B, T), ri(0, 7, B), ri(0, 24, B), ri(0, N, B)
d = Dims(N, H, Ht, ldh, ldt)
tab = Tables()
tab.E, tab.pos, tab.dur = E.data_ptr(), pos.data_ptr(), small[5].data_ptr()
for k in range(5):
    tab.time[k] = small[k].data_ptr()
bt = Batch()
bt.B, bt.T, bt.K = B, T, 0
bt.seq, bt.cw, bt.ch, bt.gap, bt.label = seq.data_ptr(), cw.data_ptr(), ch.data_ptr(), gap.data_ptr(), label.data_ptr()
for k in range(5):
    bt.pub[k] = pub[k].data_ptr()
x_icp, x_pt = torch.empty(B * T, ic, device=dev), torch.empty(B * T, pt, device=dev)
x_act, click = torch.empty(B * T, ldt, device=dev), torch.empty(B, ct, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())


def timeit(fn, iters=10):
    for _ in range(2):
        assert fn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


ms = timeit(lambda: lib.tcar_gather_clip_fwd(C.byref(d), C.byref(tab), C.byref(bt), p(x_icp), p(x_pt), p(x_act), p(click), None))
rd = B * (3536.0 * T + 512)                 # bytes read per session
wr = rd
print("gather_clip_fwd: rows=%d  %.3f ms  read %.1f GB/s  read+write %.1f GB/s  (%.0f %% / %.0f %% of 8 TB/s)" %
      (B * T, ms, rd / ms / 1e6, (rd + wr) / ms / 1e6, rd / ms / 1e6 / 80, (rd + wr) / ms / 1e6 / 80))

# ---- backward: atomics into the item gradient vs plain row output (the data-parallel mode), uniform vs skewed item ids
dx_icp, dx_pt = torch.randn(B * T, ic, device=dev) * 1e-3, torch.randn(B * T, pt, device=dev) * 1e-3
dx_act, dclick = torch.randn(B * T, ldt, device=dev) * 1e-3, torch.randn(B, ct, device=dev) * 1e-3
Gi = torch.zeros(N, ldh, device=dev)
arena = torch.zeros(40 * ldh + 150 * ldt, device=dev)
sqn = torch.zeros(64, device=dev)
rows_out = torch.empty(B * T, ldh, device=dev)


def grads(with_rows):
    gr = Grads()
    gr.g_item, gr.g_pos, gr.sqn = Gi.data_ptr(), arena.data_ptr(), sqn.data_ptr()
    off = 40 * ldh
    for k, v in enumerate((13, 32, 8, 25, 61)):
        gr.g_time[k] = arena.data_ptr() + 4 * off
        gr.slot_time[k] = 2 + k
        off += v * ldt
    gr.g_dur = arena.data_ptr() + 4 * off
    gr.slot_item, gr.slot_pos, gr.slot_dur = 0, 1, 7
    gr.rows_out = rows_out.data_ptr() if with_rows else None
    return gr


for label_, wr_ in (("atomics", False), ("rows_out", True)):
    gr = grads(wr_)
    ms = timeit(lambda: lib.tcar_gather_clip_bwd(C.byref(d), C.byref(tab), C.byref(bt), p(dx_icp), p(dx_pt), p(dx_act), p(dclick),
                                                 C.byref(gr), None))
    print("gather_clip_bwd (%s, uniform ids): rows=%d  %.1f us  %.1f ns/row" % (label_, B * T + B, ms * 1e3, ms * 1e6 / (B * T + B)))
