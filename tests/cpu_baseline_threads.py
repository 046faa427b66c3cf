"""Thread sweep of the CPU baseline (the oracle, i.e. test infrastructure: this helper lives under tests/ for that reason).
Usage (repo root): python tests/cpu_baseline_threads.py"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tcar_amd
from tcar_amd.host.synth import SynthFold
from oracle.tcar_oracle import TcarOracle, init_params_numpy
from bench import build_batches
fold = SynthFold(n_items=46033, dim=250, n_train=60000, n_test=1000, seed=2020)
rng = np.random.RandomState(2020)
batches = build_batches(fold, 4, 512, 20, rng)
params = init_params_numpy(46033, 250, 64, 0.002, 0.05, np.random.RandomState(2020))
print("cpu_count", os.cpu_count(), flush=True)
for nt in [8, 16, 32, 64, 128]:
    torch.set_num_threads(nt)
    ora = TcarOracle(params, fold.content, fold.mwdhm, dtype=torch.float32)
    ora.train_step(batches[0])
    t = time.perf_counter()
    for i in range(2): ora.train_step(batches[i + 1])
    dt = (time.perf_counter() - t) / 2
    print(nt, "threads: %.2f s/step  %.1f sessions/s" % (dt, 512 / dt), flush=True)
