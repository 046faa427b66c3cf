"""DESIGN.md §7, observation 1 — in-step repro.

ONE engine runs fused training steps; after every step (device drained) the time block of dattout that the slab-reduce kernel wrote
is recomputed on the host in fp64 from the engine's own buffers (dP, the clipped table rows, attout):
    dattout[b, ic + 64 k + c] = (sum_r dP[b, off_k + r] * tclip[off_k + r, c]) * (1 - attout[b, ic + 64 k + c]^2)
and every element off by more than rounding is attributed to the single term r whose omission explains it.
    TCAR_LIB=tools/micro/libtcar_hip_obs1.so TCAR_OBS1_LDS=1 python tools/obs1_probe.py [steps]      # round 4's LDS-staged kernel
    TCAR_LIB=tools/micro/libtcar_hip_obs1_noslp.so TCAR_OBS1_LDS=1 python tools/obs1_probe.py        # the same, built -fno-slp-vectorize
    python tools/obs1_probe.py                                                                       # the shipped (LDS-free) kernel
tools/micro/build_obs1.sh builds the two diagnostic libraries.  Other TCAR_* switches vary what runs beside the kernel."""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import tcar_amd  # noqa
from tcar_amd.engine import TcarEngine
from test_gpu_parity import _case

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N, H, Ht, B, K = 46033, 250, 64, 512, 20
params, content, mw, _ = _case(N, H, Ht, 8, 2, K, seed=61)
batches = [_case(N, H, Ht, B, T, K, seed=700 + T)[3] for T in (2, 1, 5, 3)]
eng = TcarEngine(params, content, mw, scoring="bf16x3-mixed")
res = [eng.make_resident(b) for b in batches]
OFF, NK = [0, 13, 45, 53, 78], [13, 32, 8, 25, 61]
bad_steps, bad_elems = 0, 0
by_r4, by_lane16, by_comp, by_k, by_wave = Counter(), Counter(), Counter(), Counter(), Counter()
shown = 0
for i in range(steps):
    eng.train_step(None, bt=res[i % len(res)], defer_update=True)
    torch.cuda.synchronize()
    got = eng.dattout[:B].double().cpu().numpy()
    dP = eng._dP[:B * 160].double().cpu().numpy().reshape(B, 160)
    tc = eng._tclip[:139 * 64].double().cpu().numpy().reshape(139, 64)
    att = eng.attout[:B].double().cpu().numpy()
    step_bad = 0
    for k in range(5):
        terms = dP[:, OFF[k]:OFF[k] + NK[k], None] * tc[None, OFF[k]:OFF[k] + NK[k], :]          # [B, nk, 64]
        fac = 1.0 - att[:, 512 + 64 * k:512 + 64 * (k + 1)] ** 2
        want = terms.sum(1) * fac
        g = got[:, 512 + 64 * k:512 + 64 * (k + 1)]
        tol = 2e-6 * np.abs(terms).sum(1) * np.abs(fac) + 1e-30
        bad = np.argwhere(np.abs(g - want) > tol)
        for (b, c) in bad.tolist():
            d = (want[b, c] - g[b, c]) / fac[b, c]                       # what is missing from the sum
            r = int(np.argmin(np.abs(terms[b, :, c] - d)))
            resid = abs(terms[b, r, c] - d) / (abs(d) + 1e-300)
            step_bad += 1
            by_r4[r % 4] += 1; by_comp[c % 4] += 1; by_lane16[(b % 16) % 4] += 1; by_k[k] += 1; by_wave[(b % 16) // 4] += 1
            if shown < 12:
                shown += 1
                print("step %d: row %d (tile row %d = wave %d, lane group %d) table %d col %d (component %d): missing term r = %d (r %% 4 = %d), "
                      "relative residual of that explanation %.1e" % (i, b, b % 16, (b % 16) // 4, (b % 16) % 4, k, c, c % 4, r, r % 4, resid))
    if step_bad:
        bad_steps += 1
        bad_elems += step_bad
eng.flush()
eng.check_forks()
print("obs1_probe: lib=%s lds_kernel=%s switches=%s | %d steps, %d with wrong elements, %d wrong elements" %
      (os.environ.get("TCAR_LIB", "product"), os.environ.get("TCAR_OBS1_LDS", "0"),
       {k: v for k, v in os.environ.items() if k.startswith("TCAR_") and k not in ("TCAR_LIB", "TCAR_OBS1_LDS")}, steps, bad_steps, bad_elems))
if bad_elems:
    print("  by r %% 4: %s | by float4 component: %s | by 16-lane group of the wave (rp %% 4): %s | by wave of the workgroup: %s | by table: %s" %
          (dict(by_r4), dict(by_comp), dict(by_lane16), dict(by_wave), dict(by_k)))
