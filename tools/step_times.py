"""Per-step device time of the bench's first steps (5 warm-up + 40): where does a short run lose time?
Usage: python tools/step_times.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tcar_amd  # noqa
import bench
from tcar_amd.engine import TcarEngine
from tcar_amd.host.model import initial_variables
from tcar_amd.host.synth import SynthFold

cfg = dict(bench.CONFIGS["globo"])
N, H, B, K = cfg["n_items"], cfg["hidden"], 512, 20
fold = SynthFold(n_items=N, dim=H, n_train=max(60000, 4 * B * 48), n_test=1000, seed=2020, **cfg["fold"])
batches = bench.build_batches(fold, 48, B, K, np.random.RandomState(2020), cfg)
np.random.seed(2020)
params = initial_variables(N, H, 64, 0.002, 0.05, weight_seed=2020)
eng = TcarEngine(params, fold.content, fold.mwdhm, scoring="bf16x3-mixed")
res = [eng.make_resident(b) for b in batches]
eng._ensure_work(B, max(b["seq"].shape[1] for b in batches))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(46)]
torch.cuda.synchronize()
ev[0].record()
import time
host = []
for i in range(45):
    h0 = time.perf_counter()
    eng.train_step(None, bt=res[i % len(res)], defer_update=True)
    host.append((time.perf_counter() - h0) * 1e3)
    ev[i + 1].record()
eng.flush()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(45)]
print("host ms per step:", " ".join("%.3f" % t for t in host))
print("T of the batches:", [b["seq"].shape[1] for b in batches[:45]])
print("ms per step:", " ".join("%.3f" % t for t in ts))
print("mean steps 0-4 %.4f | 5-24 %.4f | 25-44 %.4f" % (np.mean(ts[:5]), np.mean(ts[5:25]), np.mean(ts[25:45])))
