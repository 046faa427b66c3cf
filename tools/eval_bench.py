"""Evaluation-step throughput (forward + rank / top-20 + CE) at the Globo shape.  Usage: python tools/eval_bench.py [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tcar_amd  # noqa
from tcar_amd.host.synth import SynthFold
from tcar_amd.host.model import initial_variables
from tcar_amd.engine import TcarEngine
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_batches, CONFIGS
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
fold = SynthFold(n_items=46033, dim=250, n_train=60000, n_test=1000, seed=2020)
np.random.seed(2020)
eng = TcarEngine(initial_variables(46033, 250, 64, 0.002, 0.05), fold.content, fold.mwdhm, device="cuda:0", scoring="bf16x3-mixed")
batches = build_batches(fold, 16, 512, 0, np.random.RandomState(1), CONFIGS["globo"])
res = [eng.make_resident(b) for b in batches]
for i in range(5):
    eng.eval_step(None, bt=res[i % len(res)])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(iters):
    eng.eval_step(None, bt=res[i % len(res)])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print("eval step: %.3f ms  = %.0f sessions/s" % (dt * 1e3, 512 / dt))
