"""Forward / data-gradient products of the update at the bench's shapes: microseconds per call on the row-resident kernel
(k_gemm_rows, default) and on the tile kernel it replaced for them (EGP_GEMM_ROWS=0, development A/B switch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egopose_amd.gemm import gemm

def t(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

M = int(sys.argv[1]) if len(sys.argv) > 1 else 134656
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
x243, h300, h200, d52 = r(M, 243), r(M, 300), r(M, 200), r(M, 52)
W1, W2, W3 = r(300, 243), r(200, 300), r(52, 200)
b1, b2, b3 = r(300), r(200), r(52)
ctx, st = r(M, 128), r(M, 115)
idx = torch.randperm(M, device="cuda").contiguous()
xl, Wih = r(150016, 128), r(1024, 128)
cases = [
    ("fwd L1 243->300 relu", lambda: gemm(x243, W1, True, True, bias=b1, relu=True), 2 * M * 243 * 300),
    ("fwd L1 gather(128|115)->300", lambda: gemm(ctx, W1, True, True, bias=b1, relu=True, a_rows=idx, a2=st), 2 * M * 243 * 300),
    ("fwd L2 300->200 relu", lambda: gemm(h300, W2, True, True, bias=b2, relu=True), 2 * M * 300 * 200),
    ("fwd L3 200->52", lambda: gemm(h200, W3, True, True, bias=b3), 2 * M * 200 * 52),
    ("dgrad L3 52->200 mask", lambda: gemm(d52, W3, True, False, mask=h200), 2 * M * 52 * 200),
    ("dgrad L2 200->300 mask", lambda: gemm(h200, W2, True, False, mask=h300), 2 * M * 200 * 300),
    ("dgrad L1 300->128 scatter", lambda: gemm(h300, W1[:, :128], True, False, out=torch.zeros(M, 128, device="cuda"), c_rows=idx), 2 * M * 300 * 128),
    ("lstm proj 128->1024", lambda: gemm(xl, Wih, True, True), 2 * 150016 * 128 * 1024),
]
for name, fn, flop in cases:
    row = [name]
    for env in ("1", "0"):
        os.environ["EGP_GEMM_ROWS"] = env
        us = t(fn)
        row.append("%s %7.1f us %6.1f TF(6x: %5.1f%% of 2.5 PF)" % ("rows" if env == "1" else "tile", us, flop / us * 1e-6, 6 * flop / us * 1e-6 / 2500 * 100))
    print(" | ".join(row), flush=True)
os.environ.pop("EGP_GEMM_ROWS")
