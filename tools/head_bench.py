"""Micro-benchmark of the session-side forward head, op by op, at the Globo shape: each op is timed alone (HIP events around 30
launches), hot (back to back) and cold (a 256-MB device copy between launches evicts L2 / dirties the caches like the step does).
Usage: python tools/head_bench.py [T]"""
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

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, H, Ht, B, K = 46033, 250, 64, 512, 20
fold = SynthFold(n_items=N, dim=H, n_train=60000, n_test=100, seed=2020)
np.random.seed(0)
params = initial_variables(N, H, Ht, 0.002, 0.05, weight_seed=1)
eng = TcarEngine(params, fold.content, fold.mwdhm, scoring="bf16x3-mixed")
st = fold.train
idx = np.where(st.in_len == T)[0][:B]
b = st.batch_arrays(idx, "click_delta")
b["neg"] = np.random.randint(0, N, (B, K)).astype(np.int32)
bt = eng.make_resident(b)
eng.train_step(None, bt=bt)
eng.train_step(None, bt=bt)
torch.cuda.synchronize()
lib, g, p = eng.lib, eng.geo, eng._p
D = TcarEngine.desc
BT = B * T
from tcar_amd._lib import GemmDesc
junk_a = torch.empty(64 << 20, device="cuda")
junk_b = torch.empty(64 << 20, device="cuda")


def timeit(name, fn, iters=30):
    for cold in (0, 1):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        tot = 0.0
        for _ in range(iters):
            if cold:
                junk_b.copy_(junk_a)             # 256 MB through the caches, on the same stream
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        print("%-34s %s %7.1f us" % (name, "cold" if cold else "hot ", tot / iters * 1e3))


stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)     # the engine made a priority stream current: time on IT
x_c = p(eng.x_icp, g.ldh)
u = lambda k: (k + 127) // 128
stride = BT * g.ldh
sl = eng._proj_slabs


def gg(layout, descs):
    arr = (GemmDesc * len(descs))(*descs)
    assert lib.tcar_gemm_x3_grouped(layout, len(descs), arr, stream) == 0


tab = eng._tables()
timeit("gather_clip_fwd", lambda: lib.tcar_gather_clip_fwd(C.byref(eng.dims), C.byref(tab), C.byref(bt), p(eng.x_icp), p(eng.x_pt), p(eng.x_act), p(eng.click_t), stream))
unsplit = [D(BT, g.ldh, [(p(eng.x_icp), g.ic, eng._w("m_win"), g.ldh, g.ic), (x_c, g.ic, eng._w("m_wc"), g.ldh, g.ldh), (p(eng.x_act), g.ldt, eng._w("m_wint"), g.ldh, g.ldt)], p(eng.pre1), g.ldh),
           D(BT, g.ldh, [(p(eng.x_pt), g.pt, eng._w("s_win"), g.ldh, g.pt), (x_c, g.ic, eng._w("s_wc"), g.ldh, g.ldh)], p(eng.pre2), g.ldh),
           D(B, g.ldh, [(p(eng.click_t), g.ct, eng._w("q1_w"), g.ldh, g.ct)], p(eng.q1), g.ldh, bias=eng._w("q1_b"), act=1)]
timeit("proj un-split (3 problems)", lambda: gg(0, unsplit))
ps = lambda off: C.c_void_p(sl.data_ptr() + 4 * off)
split = [D(BT, g.ldh, [(p(eng.x_icp), g.ic, eng._w("m_win"), g.ldh, g.ic)], ps(0), g.ldh, splitk=u(g.ic)),
         D(BT, g.ldh, [(x_c, g.ic, eng._w("m_wc"), g.ldh, g.ldh)], ps(u(g.ic) * stride), g.ldh, splitk=u(g.ldh)),
         D(BT, g.ldh, [(p(eng.x_act), g.ldt, eng._w("m_wint"), g.ldh, g.ldt)], ps((u(g.ic) + u(g.ldh)) * stride), g.ldh, splitk=u(g.ldt)),
         D(BT, g.ldh, [(p(eng.x_pt), g.pt, eng._w("s_win"), g.ldh, g.pt)], ps(7 * stride), g.ldh, splitk=u(g.pt)),
         D(BT, g.ldh, [(x_c, g.ic, eng._w("s_wc"), g.ldh, g.ldh)], ps((7 + u(g.pt)) * stride), g.ldh, splitk=u(g.ldh)),
         D(B, g.ldh, [(p(eng.click_t), g.ct, eng._w("q1_w"), g.ldh, g.ct)], p(eng.q1), g.ldh, bias=eng._w("q1_b"), act=1)]
timeit("proj split (6 problems, 12 slabs)", lambda: gg(0, split))
timeit("proj split, first problem only", lambda: gg(0, split[:1]))
timeit("proj split, q1 only (K=128)", lambda: gg(0, split[5:]))
qd = [D(B, g.ic, [(p(eng.q1), g.ldh, eng._w("q2_w"), g.ic, g.ldh)], p(eng.q), g.ic, bias=eng._w("q2_b"), act=2)]
timeit("q = tanh(q1 Wq2 + b)  K=256", lambda: gg(0, qd))
timeit("attn_pool_fwd (plain)", lambda: lib.tcar_attn_pool_fwd(C.byref(eng.dims), B, T, p(eng.x_icp), p(eng.x_pt), p(eng.pre1), p(eng.pre2), p(eng.q), eng._w("m_wres"), eng._w("s_wres"), p(eng.pooled), p(eng.alpha), stream))
timeit("attn_pool_fwd_slabs (7 + 5)", lambda: lib.tcar_attn_pool_fwd_slabs(C.byref(eng.dims), B, T, p(eng.x_icp), p(eng.x_pt), ps(0), 7, ps(7 * stride), 5, stride, p(eng.pre1), p(eng.pre2), p(eng.q), eng._w("m_wres"), eng._w("s_wres"), p(eng.pooled), p(eng.alpha), stream))
att = [D(B, g.ic, [(p(eng.pooled), g.ek, eng._w("o_w"), g.ic, g.ic)], p(eng.attout), g.ek, bias=eng._w("o_b"), act=2),
       D(B, g.pt, [(p(eng.pooled, g.ic), g.ek, eng._w("ot_w"), g.pt, g.pt)], p(eng.attout, g.ic), g.ek, bias=eng._w("ot_b"), act=2)]
timeit("attout (K=512 / 320, no planes)", lambda: gg(0, att))
# an empty-ish kernel for the floor: zero 64 floats
z = torch.zeros(64, device="cuda")
timeit("torch zero_ of 64 floats (floor)", lambda: z.zero_())
