"""The order-fixed small-table backward (tcar_small_tables_bwd_det) ALONE against the bucket length T, uniform ids and the
skew of the benchmark's synthetic fold (every session in ONE month: one table row collects every source).
Usage: python tools/small_det_bench.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import tcar_amd  # noqa
from tcar_amd import _lib, torch_ops
from tcar_amd._lib import Grads
from test_gpu_torch_ops import _tables, _feed

DEV = "cuda:0"
lib = _lib.load()
N, H, Ht, B = 3000, 250, 64, 512
for T in (1, 2, 4, 7, 10, 40):
    for skew in (False, True):
        rng = np.random.RandomState(T)
        E, pos, small, ldh, ldt, ek = _tables(N, H, Ht, rng, scale=0.2)
        seq, pub, gap, cw, ch, feed = _feed(B, T, N, rng)
        n = B * T
        if skew:
            feed[n:2 * n] = 5                                        # every publish time in one month
        Et, post, smt, fd = (torch.tensor(a, device=DEV) for a in (E, pos, small, feed))
        mk = lambda *s: torch.tensor(rng.standard_normal(s).astype(np.float32), device=DEV)
        dx_icp, dx_pt, dx_act, dclick = mk(n, 2 * ldh), mk(n, 5 * ldt), mk(n, ldt), mk(B, 2 * ldt)
        dims, _, _, _ = torch_ops._geom(Et, post, smt, H, Ht)
        tab, bt = torch_ops._tables(Et, post, smt, ldt), torch_ops._batch(fd, B, T)
        p = lambda t: C.c_void_p(t.data_ptr())
        g_item, g_pos = torch.zeros(N, ldh, device=DEV), torch.zeros(40, ldh, device=DEV)
        g_small, sqn = torch.zeros(150, ldt, device=DEV), torch.zeros(_lib.NSLOT, device=DEV)
        gr = Grads()
        gr.g_item, gr.g_pos, gr.sqn = g_item.data_ptr(), g_pos.data_ptr(), sqn.data_ptr()
        for k in range(5):
            gr.g_time[k] = g_small.data_ptr() + 4 * torch_ops._ROWOFF[k] * ldt
            gr.slot_time[k] = 2 + k
        gr.g_dur = g_small.data_ptr() + 4 * torch_ops._ROWOFF[5] * ldt
        gr.slot_item, gr.slot_pos, gr.slot_dur = 0, 1, 7
        ws = torch.zeros(lib.tcar_small_det_ws_floats(), device=DEV)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        run = lambda: lib.tcar_small_tables_bwd_det(C.byref(dims), C.byref(tab), C.byref(bt), p(dx_icp), p(dx_pt), p(dx_act), p(dclick),
                                                    C.byref(gr), p(ws), st)
        for _ in range(5):
            assert run() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        print("T = %2d  %s  %.1f us per call (two launches: the table pass + the norm fold)" % (T, "one month" if skew else "uniform  ", e0.elapsed_time(e1) / 50 * 1e3))
