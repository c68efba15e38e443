"""egp_gemm_f32 (bf16 matrix cores, split operands) against the library float32 products at the update's shapes.
Usage: gemm_x3_probe.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egopose_amd.gemm import gemm, linear_fwd, linear_dgrad, linear_wgrad

n = int(sys.argv[1]) if len(sys.argv) > 1 else 139264
dev = "cuda"


def bench(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = {"lib": 0.0, "x6": 0.0, "x3": 0.0, "x1": 0.0}
for (k, m) in [(243, 300), (300, 200), (200, 52), (200, 1), (128, 1024)]:
    x = torch.randn(n, k, device=dev); w = torch.randn(m, k, device=dev) * 0.1; b = torch.randn(m, device=dev)
    gy = torch.randn(n, m, device=dev); h = torch.randn(n, k, device=dev)
    fl = 2.0 * n * k * m
    rows = []
    for name, lib, f in [
        ("fwd+bias+relu", lambda: torch.relu(torch.addmm(b, x, w.t())), lambda t: (lambda: linear_fwd(x, w, b, True, terms=t))),
        ("dgrad*mask", lambda: (gy @ w) * (h > 0), lambda t: (lambda: linear_dgrad(gy, w, mask=h, terms=t))),
        ("wgrad+bias", lambda: (gy.t() @ x, gy.sum(0)), lambda t: (lambda: linear_wgrad(gy, x, terms=t))),
    ]:
        tl, t6, t3, t1 = bench(lib), bench(f(6)), bench(f(3)), bench(f(1))
        tot["lib"] += tl; tot["x6"] += t6; tot["x3"] += t3; tot["x1"] += t1
        print("%4dx%4d %-14s library f32 %7.1f us (%5.1f TF/s) | 3 pieces (x6) %7.1f us (%6.1f TF/s eff) | 2 pieces (x3) %7.1f us | bf16 (x1) %7.1f us" % (
            k, m, name, tl, fl / tl / 1e6, t6, fl / t6 / 1e6, t3, t1))
print("sum over shapes (us):", {k: round(v) for k, v in tot.items()})
