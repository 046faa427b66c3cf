"""Diagnostic (follow-up of tools/thread_probe.py): which buffer of engine 0's FIRST training step differs when another engine is
stepped from a second host thread at the same time?  Engine 0 is rebuilt for every trial (same initial state), runs ONE step, and
its forward / backward buffers are compared with the solo run of the same step."""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["TCAR_NO_PRIO"] = "1"
import numpy as np
import torch
import tcar_amd  # noqa
from tcar_amd.engine import TcarEngine
from test_gpu_parity import _case

mask = int(sys.argv[1]) if len(sys.argv) > 1 else -1
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 12
H, Ht, K = 250, 64, 20
case0, case1 = _case(46033, H, Ht, 512, 2, K, seed=61), _case(9000, H, Ht, 256, 3, K, seed=62)
NAMES = ["x_icp", "x_pt", "click_t", "q", "pooled", "attout", "a16h", "a16l", "_p16h", "_p16l", "ce", "neg_fb", "loss", "dl16h", "dattout",
         "dpooled", "dq", "dq1", "dclick", "dx_icp", "dx_pt", "Gi", "G", "sqn_dense"]


def one_step(stream, go=None):
    params, content, mw, batch = case0
    with torch.cuda.stream(stream):
        eng = TcarEngine(params, content, mw, scoring="bf16x3-mixed")
        if mask >= 0:
            eng.set_tuning(TCAR_FLAG_FORK=mask)
        bt = eng.make_resident(batch)
        eng._ensure_work(bt.B, bt.T)
        eng._ctx()
        torch.cuda.synchronize()
        if go is not None:
            go.set()
        eng.train_step(None, bt=bt, defer_update=True)
        torch.cuda.synchronize()
        out = {}
        for n in NAMES:
            t = getattr(eng, n, None)
            if t is not None:
                out[n] = t.detach().float().cpu().numpy().copy()
        eng.flush()
    return out


solo = one_step(torch.cuda.Stream(priority=-1))
again = one_step(torch.cuda.Stream(priority=-1))
print("solo repeatable:", all(np.array_equal(solo[k], again[k], equal_nan=True) for k in solo))
stop = threading.Event()
started = threading.Event()


def hammer():
    params, content, mw, batch = case1
    with torch.cuda.stream(torch.cuda.Stream(priority=-1)):
        eng = TcarEngine(params, content, mw, scoring="bf16x3-mixed")
        bt = eng.make_resident(batch)
        started.set()
        while not stop.is_set():
            for _ in range(20):
                eng.train_step(None, bt=bt, defer_update=True)
            torch.cuda.synchronize()
        eng.flush()


th = threading.Thread(target=hammer)
th.start()
started.wait()
for trial in range(trials):
    res = {}
    t = threading.Thread(target=lambda: res.update(one_step(torch.cuda.Stream(priority=-1))))
    t.start()
    t.join()
    bad = []
    for k in NAMES:
        if k in solo and not np.array_equal(solo[k], res[k], equal_nan=True):
            d = np.abs(solo[k].astype(np.float64) - res[k])
            rows = np.unique(np.where(d.reshape(d.shape[0], -1).max(1) > 0)[0]) if d.ndim > 1 else np.where(d > 0)[0]
            bad.append("%s(%d rows, first %s, max %.2e)" % (k, len(rows), rows[:6].tolist(), d.max()))
    print("trial %d: %s" % (trial, "identical" if not bad else " ".join(bad)))
stop.set()
th.join()
