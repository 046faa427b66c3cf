"""Would a hipGraph replay of the fused training step be faster than the eager C++ driver?  Captures one step per resident
batch (tcar_graph_probe) and times the replays; prints eager vs graph ms per step.  Usage: python tools/graph_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tcar_amd  # noqa
from tcar_amd.engine import TcarEngine
from tcar_amd.host.model import initial_variables
from tcar_amd.host.synth import SynthFold

N, H, B, K = 46033, 250, 512, 20
fold = SynthFold(n_items=N, dim=H, n_train=60000, n_test=10, seed=2020)
np.random.seed(2020)
params = initial_variables(N, H, 64, 0.002, 0.05, weight_seed=2020)
eng = TcarEngine(params, fold.content, fold.mwdhm, scoring="bf16x3-mixed")
rng = np.random.RandomState(1)
res = []
for T in (1, 2, 3, 5):
    idx = np.where(fold.train.in_len == T)[0][:B]
    b = fold.train.batch_arrays(idx, "click_delta")
    b["neg"] = rng.randint(0, N, size=(len(idx), K)).astype(np.int32)
    res.append((T, eng.make_resident(b)))
eng._ensure_work(B, 5)
for T, bt in res:
    for _ in range(5):
        eng.train_step(None, bt=bt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        eng.train_step(None, bt=bt)
    e1.record()
    torch.cuda.synchronize()
    eager = e0.elapsed_time(e1) / 100
    ms = C.c_float(0)
    rc = eng.lib.tcar_graph_probe(C.byref(eng._ctx()), C.byref(bt), C.c_float(1e-3), 100, C.byref(ms), eng._stream())
    print("T=%d  eager %.4f ms/step   graph replay %s" % (T, eager, ("%.4f ms/step" % ms.value) if rc == 0 else "failed rc=%d" % rc))
