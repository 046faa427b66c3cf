"""Gradient error of the FUSED training step against the fp64 oracle at the benched size, for the anchored softmax form
(TCAR_FUSED_CE = 2) and the group-maximum form with the rescale pass (1): norm-wise relative error of all 23 gradients, the clip
norms, and the distance between the two forms.  python tools/anchor_grad_error.py [T]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parity import _case, rel_norm
from oracle.tcar_oracle import TcarOracle
from tcar_amd.engine import TcarEngine

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, H, Ht, B, K = 46033, 250, 64, 512, 20
params, content, mw, batch = _case(N, H, Ht, B, T, K, seed=1000 + T, emb_std=0.05, w_std=0.05)
ora = TcarOracle(params, content, mw)
o, g_o, sq_o = ora.loss_and_grads(batch)
g_o = {k: v.numpy() for k, v in g_o.items()}
res = {}
for f in (2, 1):
    eng = TcarEngine(params, content, mw, scoring="bf16x3-mixed")
    eng.set_tuning(TCAR_FUSED_CE=f)
    bt = eng.make_resident(batch)
    print("form", f, eng.step_form(bt))
    loss = eng.train_step(None, bt=bt, defer_update=True)
    torch.cuda.synchronize()
    res[f] = (eng.export_grads(), eng.export_sqnorms(), loss.cpu().numpy().copy())
    del eng
print("loss rel err: anchored %.3g  rescaled %.3g" % (rel_norm(res[2][2][:B], o["loss"].detach().numpy()), rel_norm(res[1][2][:B], o["loss"].detach().numpy())))
print("%-14s %12s %12s %12s   %s" % ("variable", "anchored", "rescaled", "between", "sqnorm rel err (anchored, rescaled)"))
for k in g_o:
    a, r = res[2][0][k], res[1][0][k]
    print("%-14s %12.3e %12.3e %12.3e   %.2e %.2e" % (k, rel_norm(a, g_o[k]), rel_norm(r, g_o[k]), rel_norm(a, r),
          abs(res[2][1][k] - sq_o[k]) / max(sq_o[k], 1e-300), abs(res[1][1][k] - sq_o[k]) / max(sq_o[k], 1e-300)))
