"""Micro-benchmark of tcar_softmax_ce_bf16 at the Globo shape.  Usage: python tools/softmax_bench.py [iters]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tcar_amd  # noqa
from tcar_amd import _lib
lib = _lib.load()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B, N = 512, int(os.environ.get("GB_N", 46033))
Npad = (N + 127) // 128 * 128
p = lambda t: C.c_void_p(t.data_ptr())
logits = torch.randn(B, Npad, device="cuda") * 3
label = torch.randint(0, N, (B,), dtype=torch.int32, device="cuda")
ce = torch.empty(B, device="cuda")
dh = torch.empty(B, Npad, dtype=torch.bfloat16, device="cuda"); dl = torch.empty_like(dh)
def run(): assert lib.tcar_softmax_ce_bf16(B, N, p(logits), Npad, p(label), p(ce), p(dh), p(dl), None) == 0
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
ref = torch.logsumexp(logits[:, :N].double(), 1) - logits[torch.arange(B), label.long()].double()
print("softmax (row-resident form): %.1f us  (read+write %.0f GB/s)  ce err %.2e" % (ms * 1e3, B * Npad * 8 / ms / 1e6, float((ce.double() - ref).abs().max())))
