"""Micro-benchmark of the split-bf16 scoring GEMMs at the Globo shape (B=512, N=46033, K=832).
Usage: python tools/gemm_bench.py [fwd|fwdce|fwdce2|dx|de|dx2|de2|both|both2] [nsplit] [iters]
(fwdce: the logits GEMM with its softmax epilogue; fwdce2: + one-hot time segment; dx2 / de2: the one-hot form of the gradient GEMMs,
 hi planes only — tcar_gemm_bf16_dx_onehot, tcar_gemm_bf16_de_qz; both2: dx2 and de2 concurrently)
`both`: dX on a high-priority stream and dE on a second stream, concurrently, as the step driver runs them."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tcar_amd  # noqa
from tcar_amd import _lib

lib = _lib.load()
torch.manual_seed(1234)
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
nsplit = int(sys.argv[2]) if len(sys.argv) > 2 else 3
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
N = int(os.environ.get("GB_N", 46033))
SK = int(os.environ.get("GB_SPLITK", 18))
B, Npad, EK = 512, (N + 127) // 128 * 128, 832
bf = dict(dtype=torch.bfloat16, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
e_h, e_l = torch.randn(Npad, EK, **bf), torch.randn(Npad, EK, **bf) * 0.004
a_h, a_l = torch.randn(B, EK, **bf), torch.randn(B, EK, **bf) * 0.004
a_s = (a_h.float() * 0.1).to(torch.bfloat16)
d_h, d_l = torch.randn(B, Npad, **bf) * 0.01, torch.randn(B, Npad, **bf) * 1e-4
ap_h, ap_l = torch.randn(B, 576, **bf), torch.randn(B, 576, **bf) * 0.004
logits = torch.empty(B, Npad, device="cuda")
slabs = torch.empty(SK, B, EK, device="cuda")
gi, det = torch.empty(N, 256, device="cuda"), torch.empty(N, 320, device="cuda")
plane = torch.zeros(B, Npad, **bf)
nstat = B * ((N + 63) // 64 + 8) * 2
stats, lab_logit = torch.empty(nstat, device="cuda"), torch.empty(B, device="cuda")
label = torch.randint(0, N, (B,), dtype=torch.int32, device="cuda")
gw, ng = C.c_int32(0), C.c_int32(0)
p_h, p_l = torch.randn(B, 160, **bf), torch.randn(B, 160, **bf) * 0.004
oh = (torch.rand(Npad, 160, device="cuda") < 5.0 / 160).to(torch.bfloat16)


mw = torch.stack([torch.randint(0, v, (N,), device="cuda") for v in (13, 32, 8, 25, 61)], 1).to(torch.int32).contiguous()
rows = (mw.long() + torch.tensor([0, 13, 45, 53, 78], device="cuda")).t().reshape(-1)
order = torch.argsort(rows, stable=True)
et_perm = torch.empty(5 * N, dtype=torch.int32, device="cuda")
et_perm[order] = torch.arange(5 * N, dtype=torch.int32, device="cuda")
tclip = torch.randn(160 * 64 + 320, device="cuda") * 0.1
qz = torch.empty(5 * N, 2, device="cuda")
slabs2 = torch.empty(SK, B, 672, device="cuda")
TILE = int(os.environ.get("GB_TILE", 0))
hp, s2 = torch.cuda.Stream(priority=-1), torch.cuda.Stream()


def run_both2():
    cur = torch.cuda.current_stream()
    hp.wait_stream(cur)
    s2.wait_stream(cur)
    rc1 = lib.tcar_gemm_bf16_de_qz(N, B, p(d_h), Npad, B, p(ap_h), 576, B, 256, p(gi), 256, p(mw), p(et_perm), p(tclip), p(qz), TILE,
                                   C.c_void_p(s2.cuda_stream))
    rc0 = lib.tcar_gemm_bf16_dx_onehot(B, 512, Npad, p(d_h), Npad, B, p(e_h), EK, Npad, p(oh), 160, p(slabs2), 672, SK,
                                       C.c_void_p(hp.cuda_stream))
    cur.wait_stream(hp)
    cur.wait_stream(s2)
    return rc0 or rc1, 2.0 * B * N * (820 + 570)


def run_both():
    """dX on the priority stream, dE on the other, joined before the next iteration (like backward_impl)"""
    cur = torch.cuda.current_stream()
    hp.wait_stream(cur)
    s2.wait_stream(cur)
    rc1 = lib.tcar_gemm_bf16(2, N, 576, B, p(d_h), p(d_l), Npad, B, p(ap_h), p(ap_l), 576, B, p(gi), 256, p(det), 320, 256, nsplit, 1,
                             C.c_void_p(s2.cuda_stream))
    rc0 = lib.tcar_gemm_bf16(0, B, EK, Npad, p(d_h), p(d_l), Npad, B, p(e_h), p(e_l), EK, Npad, p(slabs), EK, None, 0, 0, nsplit, SK,
                             C.c_void_p(hp.cuda_stream))
    cur.wait_stream(hp)
    cur.wait_stream(s2)
    return rc0 or rc1, 2.0 * B * N * (820 + 570)


def run():
    if which == "both":
        return run_both()
    if which == "both2":
        return run_both2()
    if which == "dx2":
        return lib.tcar_gemm_bf16_dx_onehot(B, 512, Npad, p(d_h), Npad, B, p(e_h), EK, Npad, p(oh), 160, p(slabs2), 672, SK, None), 2.0 * B * N * 820
    if which == "de2":
        return lib.tcar_gemm_bf16_de_qz(N, B, p(d_h), Npad, B, p(ap_h), 576, B, 256, p(gi), 256, p(mw), p(et_perm), p(tclip), p(qz), TILE,
                                        None), 2.0 * B * N * 570
    if which == "fwd":
        return lib.tcar_gemm_bf16(1, B, N, EK, p(a_h), p(a_l), EK, B, p(e_h), p(e_l), EK, Npad, p(logits), Npad, None, 0, 0, nsplit, 1, None), 2.0 * B * N * 820
    if which == "fwdce":
        return lib.tcar_gemm_bf16_ce(B, N, EK, p(a_h), p(a_l), EK, B, p(e_h), p(e_l), EK, Npad, EK, None, None, None, 0, p(plane), Npad, B,
                                     p(stats), nstat, p(label), p(lab_logit), nsplit, C.byref(gw), C.byref(ng), None), 2.0 * B * N * 820
    if which == "fwdce2":   # item | content columns + the 160-column one-hot segment of the publish-time rows; the step's ANCHORED
        # epilogue (round 6: exp(accumulator), no group maxima — attout scaled so that the synthetic logits stay inside exp's range)
        return lib.tcar_gemm_bf16_ce_anchor(B, N, 512 + 160, p(a_s), p(a_l), EK, B, p(e_h), p(e_l), EK, Npad, 512, p(p_h), p(p_l), p(oh), 160,
                                            p(plane), Npad, B, p(stats), nstat, p(label), p(lab_logit), nsplit, C.byref(gw), C.byref(ng),
                                            None), 2.0 * B * N * 820
    if which == "dx":
        return lib.tcar_gemm_bf16(0, B, EK, Npad, p(d_h), p(d_l), Npad, B, p(e_h), p(e_l), EK, Npad, p(slabs), EK, None, 0, 0, nsplit, SK, None), 2.0 * B * N * 820
    return lib.tcar_gemm_bf16(2, N, 576, B, p(d_h), p(d_l), Npad, B, p(ap_h), p(ap_l), 576, B, p(gi), 256, p(det), 320, 256, nsplit, 1, None), 2.0 * B * N * 570


for _ in range(3):
    rc, fl = run()
    assert rc == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
if os.environ.get("GB_SUM") and which.startswith("fwdce"):    # checksum of the outputs: variants of one kernel must agree bit for bit
    ng_, gw_ = ng.value, gw.value
    st = stats[: B * ng_ * 2].view(B, ng_, 2)
    print("checksum plane %.10e stats %.10e lab %.10e tile=%s " % (plane.double().abs().sum().item(), st.double().sum().item(),
                                                                lab_logit.double().sum().item(), os.environ.get("TCAR_BF16_TILE", "0")), end="")
if os.environ.get("GB_SUM") and which in ("de2", "both2"):
    print("checksum gi %.10e qz %.10e " % (gi.double().abs().sum().item(), qz.double().abs().sum().item()), end="")
if os.environ.get("GB_SUM") and which in ("dx2", "both2"):
    print("checksum slabs %.10e abs %.10e " % (slabs2.double().sum().item(), slabs2.double().abs().sum().item()), end="")
print("N=%d sk=%d " % (N, SK), end="")
print("%s nsplit=%d: %.1f us  alg %.0f TF  executed %.0f TF" % (which, nsplit, ms * 1e3, fl / ms / 1e9, fl * (3 if nsplit == 3 else 1) / ms / 1e9))
