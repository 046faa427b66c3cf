"""Where the host time of one training epoch goes (sampler / upload / step enqueue) next to the device time.
Usage: python tools/host_profile.py [n_train]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tcar_amd  # noqa
from tcar_amd.host.synth import SynthFold
from tcar_amd.host.sampler import Sampler
from tcar_amd.host.model import initial_variables
from tcar_amd.engine import TcarEngine

n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
fold = SynthFold(n_items=46033, dim=250, n_train=n_train, n_test=1000, seed=2020)
store = fold.train
len_d = {int(T): np.where(store.in_len == T)[0].tolist() for T in np.unique(store.in_len)}
np.random.seed(2020)
eng = TcarEngine(initial_variables(46033, 250, 64, 0.002, 0.05), fold.content, fold.mwdhm, device="cuda:0", scoring="bf16x3")
for ep in range(2):
    t_init = time.perf_counter()
    s = Sampler(len_d, None, None, {0: [0]}, fold.item_dict, 20, batch_size=512, gap_mode="active_t", neg_mode="uniform", store=store)
    t_init = time.perf_counter() - t_init
    ts = tu = tl = 0.0
    n = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while s.has_next():
        a = time.perf_counter()
        f = s.next_batch_arrays()
        b = time.perf_counter()
        bt = eng.upload(f)
        c = time.perf_counter()
        eng.train_step(None, bt=bt)
        d = time.perf_counter()
        ts += b - a; tu += c - b; tl += d - c; n += 1
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("epoch %d: %d batches; sampler init %.1f ms (%.3f ms/batch); per batch: sampler %.3f, upload %.3f, step enqueue %.3f ms; "
          "host loop %.3f ms/batch, with device drain %.3f ms/batch" % (ep, n, t_init * 1e3, t_init / n * 1e3, ts / n * 1e3, tu / n * 1e3,
                                                                       tl / n * 1e3, t_host / n * 1e3, t_all / n * 1e3))
