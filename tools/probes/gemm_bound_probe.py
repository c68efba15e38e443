"""The update's product shapes alone, on random operands (no rollout, no indices: safe for the timing-only builds of egp_gemm.hip --
EGP_WS_SKIP=1 consumers idle, =2 producers do not stage, EGP_FAKE_SPLIT=2 no operand split -- whose numerics are garbage).
Inputs rotate over 3 sets so that nothing is re-read from the Infinity Cache. us per call, float32-equivalent TFLOP/s, TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egopose_amd import gemm as G

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
torch.manual_seed(0)
n = 130964
R = 3


def rnd(*shape):
    return [torch.randn(*shape, device=dev) for _ in range(R)]


x243, x300, x200 = rnd(n, 243), rnd(n, 300), rnd(n, 200)
w1, w2 = torch.randn(300, 243, device=dev) * 0.05, torch.randn(200, 300, device=dev) * 0.05
b1, b2 = torch.zeros(300, device=dev), torch.zeros(200, device=dev)
kl = 281600
dpre, xl = rnd(kl, 512), rnd(kl, 128)
cases = [
    ("fwd   131k x 300 x 243  kc kc bias relu", lambda i: G.gemm(x243[i], w1, True, True, bias=b1, relu=True), n, 300, 243),
    ("fwd   131k x 200 x 300  kc kc bias relu", lambda i: G.gemm(x300[i], w2, True, True, bias=b2, relu=True), n, 200, 300),
    ("dgrad 131k x 300 x 200  kc rc mask", lambda i: G.gemm(x200[i], w2, True, False, mask=x300[i]), n, 300, 200),
    ("wgrad 300 x 243 x 131k  rc rc splits 42 + bias", lambda i: G.gemm(x300[i], x243[i], False, False, splits=42, want_bias_grad=True), 300, 243, n),
    ("wgrad 200 x 300 x 131k  rc rc splits 42 + bias", lambda i: G.gemm(x200[i], x300[i], False, False, splits=42, want_bias_grad=True), 200, 300, n),
    ("wgrad 512 x 128 x 282k  rc rc splits 64", lambda i: G.gemm(dpre[i], xl[i], False, False, splits=64), 512, 128, kl),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
if os.environ.get("ONLY"):                      # ONLY=<case index>: one shape (for counter passes)
    cases = [cases[int(os.environ["ONLY"])]]
tot = 0.0
for name, fn, M, N, K in cases:
    for i in range(R):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        fn(r % R)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    tot += us
    nbytes = 4.0 * (M * K + N * K + M * N)
    print("%8.1f us  %6.1f TF/s  %5.2f TB/s  %s" % (us, 2.0 * M * N * K / us / 1e6, nbytes / us / 1e6, name))
print("sum %.1f us" % tot)
