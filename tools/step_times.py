"""Per-step device time of the bench's first steps (5 warm-up + 40): where does a short run lose time?
Usage: python tools/step_times.py            pre-formed feeds, 45 steps in one loop
       python tools/step_times.py sampler    the driver's form: 5 warm-up steps + flush + sync, then 20 steps, device sampler in the loop"""
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
batches, batch_ids = bench.build_batches(fold, 48, B, K, np.random.RandomState(2020), cfg, with_ids=True)
np.random.seed(2020)
params = initial_variables(N, H, 64, 0.002, 0.05, weight_seed=2020)
eng = TcarEngine(params, fold.content, fold.mwdhm, scoring="bf16x3-mixed")
res = [eng.make_resident(b) for b in batches]
eng._ensure_work(B, max(b["seq"].shape[1] for b in batches))
import time
if len(sys.argv) > 1 and sys.argv[1] == "sampler":
    from tcar_amd.device_sampler import DeviceSampler
    ds = DeviceSampler(eng, fold.train, cfg["neg_mode"], None, fold.item_dict, seed=2020)
    for rep in range(3):
        ds.plan(batch_ids[0:5])
        for bt in ds.planned(K, cfg["gap_mode"]):
            eng.train_step(None, bt=bt, defer_update=True)
        eng.flush()
        ds.plan(batch_ids[5:25])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(22)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        host, gen = [], []
        it = ds.planned(K, cfg["gap_mode"])
        for i in range(20):
            g0 = time.perf_counter()
            bt = next(it)
            g1 = time.perf_counter()
            eng.train_step(None, bt=bt, defer_update=True)
            host.append((time.perf_counter() - g1) * 1e3)
            gen.append((g1 - g0) * 1e3)
            ev[i + 1].record()
        for _ in it:
            pass
        eng.flush()
        ev[21].record()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(21)]
        print("run %d: %.4f ms/step wall (enqueue %.4f), T = %s" % (rep, dt * 1e3 / 20, t_enq * 1e3 / 20, [b["seq"].shape[1] for b in batches[5:25]]))
        print("  sampler next() ms:", " ".join("%.3f" % t for t in gen))
        print("  train_step host ms:", " ".join("%.3f" % t for t in host))
        print("  device ms per step:", " ".join("%.3f" % t for t in ts[:20]), "| flush %.3f" % ts[20])
    sys.exit(0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(46)]
torch.cuda.synchronize()
ev[0].record()
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
