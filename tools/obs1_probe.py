"""DESIGN.md §7, observation 1 — in-step repro attempt.

Two engines built from the same variables step the same batches alternately (the device is drained between steps); after every
step their dattout, dP and slab buffers are compared bit for bit.  The fused step is order-fixed, so ANY difference is the
observation.  Run with the diagnostic library (tools/micro/build_obs1.sh) to put round 4's LDS-staged reduce kernel back:
    TCAR_LIB=tools/micro/libtcar_hip_obs1.so TCAR_OBS1_LDS=1 python tools/obs1_probe.py [steps]
and without TCAR_OBS1_LDS as the control (the shipped LDS-free kernel of the same binary).  Other TCAR_* switches vary what runs
beside the kernel (TCAR_BF16_TILE=256: dE at one 12-wave workgroup per CU; TCAR_FLAG_FORK=0: event forks only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import tcar_amd  # noqa
from tcar_amd.engine import TcarEngine
from test_gpu_parity import _case

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N, H, Ht, B, K = 46033, 250, 64, 512, 20
params, content, mw, _ = _case(N, H, Ht, 8, 2, K, seed=61)
batches = [_case(N, H, Ht, B, T, K, seed=700 + T)[3] for T in (2, 1, 5, 3)]
engs = [TcarEngine(params, content, mw, scoring="bf16x3-mixed") for _ in range(2)]
res = [[e.make_resident(b) for b in batches] for e in engs]
names = ["dattout", "_dP", "loss"]
bad_steps, total_bad = 0, 0
first = None
for i in range(steps):
    outs = []
    for e, r in zip(engs, res):
        e.train_step(None, bt=r[i % len(r)], defer_update=True)
        torch.cuda.synchronize()
        T = batches[i % len(batches)]["seq"].shape[1]
        outs.append({n: getattr(e, n).detach().float().cpu().numpy().copy() for n in names if getattr(e, n, None) is not None})
    diff = {}
    for n in outs[0]:
        a, b = outs[0][n], outs[1][n]
        if not np.array_equal(a, b, equal_nan=True):
            idx = np.argwhere(a != b)
            diff[n] = (len(idx), idx[:4].tolist())
    if diff:
        bad_steps += 1
        total_bad += sum(v[0] for v in diff.values())
        if first is None:
            first = (i, diff)
        if bad_steps <= 5:
            print("step %d: %s" % (i, diff))
for e in engs:
    e.flush()
    e.check_forks()
print("obs1_probe: lib=%s lds_kernel=%s switches=%s | %d steps, %d steps with differences, %d differing elements; first: %s"
      % (os.environ.get("TCAR_LIB", "product"), os.environ.get("TCAR_OBS1_LDS", "0"),
         {k: v for k, v in os.environ.items() if k.startswith("TCAR_") and k not in ("TCAR_LIB", "TCAR_OBS1_LDS")}, steps, bad_steps, total_bad, first))
