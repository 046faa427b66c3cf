"""Micro-benchmark of the small split-bf16 GEMM (tcar_gemm_x3_grouped) on the step's shapes, alone on the chip: K sweep
(slope = time per 64-deep stage, intercept = fixed cost) for every ring depth / stage depth.
Usage: python tools/x3_bench.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tcar_amd  # noqa
from tcar_amd import _lib
from tcar_amd._lib import GemmDesc

lib = _lib.load()
p = lambda t: t.data_ptr()


def desc(M, N, segs, Cm, ldc, bias=None, act=0):
    d = GemmDesc()
    d.nseg = len(segs)
    for i, (A, lda, Bm, ldb, K) in enumerate(segs):
        d.A[i], d.lda[i], d.B[i], d.ldb[i], d.K[i] = p(A), lda, p(Bm), ldb, K
    d.C, d.ldc, d.bias = p(Cm), ldc, (p(bias) if bias is not None else None)
    d.M, d.N, d.act, d.beta, d.splitk, d.atomic = M, N, act, 0, 1, 0
    return d


def timeit(layout, descs, iters=50):
    arr = (GemmDesc * len(descs))(*descs)
    for _ in range(3):
        assert lib.tcar_gemm_x3_grouped(layout, len(descs), arr, None) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.tcar_gemm_x3_grouped(layout, len(descs), arr, None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


dev = "cuda"
M, N = 1100, 256
X = torch.randn(M, 2048, device=dev)
W = torch.randn(2048, N, device=dev) * 0.05
Wt = torch.randn(512, 2048, device=dev) * 0.05
Y = torch.empty(M, 512, device=dev)
bias = torch.zeros(512, device=dev)
for ring, xk in ((1, 64),):      # (the ring-depth / stage-depth sweep of round 2 is in profiles/r02_x3_small_gemm_bench.txt)
    row = []
    for K in (64, 128, 256, 512, 1024, 2048):
        row.append("K=%d %.1f" % (K, timeit(0, [desc(M, N, [(X, 2048, W, N, K)], Y, 512)])))
    seg = timeit(0, [desc(M, N, [(X, 2048, W, N, 512), (X, 2048, W, N, 256), (X, 2048, W, N, 64)], Y, 512),
                     desc(M, N, [(X, 2048, W, N, 320), (X, 2048, W, N, 256)], Y[:, 256:], 512),
                     desc(512, N, [(X, 2048, W, N, 128)], Y, 512, bias=bias, act=1)])
    q = timeit(0, [desc(512, 512, [(X, 2048, Wt.t().contiguous()[:256], 512, 256)], Y, 512, bias=bias, act=2)])
    bw = timeit(1, [desc(512, 512, [(X, 2048, Wt, 2048, 512)], Y, 512), desc(512, 320, [(X, 2048, Wt, 2048, 320)], Y, 512)])
    print("ring=%d xk=%d  NN M=1100 N=256: %s | fwd-proj group %.1f | q %.1f | bwd NT 512x512x512 + 512x320x320 %.1f us"
          % (ring, xk or 64, "  ".join(row), seg, q, bw))
