import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import tcar_amd
from test_gpu_parity import _case
from oracle.tcar_oracle import TcarOracle
from tcar_amd.engine import TcarEngine
N, H, Ht, B, T, K = 1000, 250, 64, 64, 4, 20
params, content, mw, batch = _case(N, H, Ht, B, T, K, seed=78)
ora = TcarOracle(params, content, mw)
o, g_o, _ = ora.loss_and_grads(batch)
for m in ("f32", "bf16x3", "bf16x3-mixed"):
    eng = TcarEngine(params, content, mw, scoring=m)
    loss = eng.loss_and_grads(batch).cpu().numpy()
    d = np.abs(loss - o["loss"].detach().numpy())
    print(m, d.max(), d.argmax(), loss[34], float(o["loss"][34]))
