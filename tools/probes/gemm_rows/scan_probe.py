"""Black-box timing of the row-resident kernel (gemm_rows_r5.patch applied): per-step cost and fixed cost per tile from a scan over K, at
M = whole rounds of the chip (512 row tiles of 128 rows per round with 4-wave workgroups, 256 tiles of 256 rows with 8-wave ones)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from egopose_amd.gemm import gemm

def t(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

g = torch.Generator(device="cuda").manual_seed(0)
for N in (320, 128):
    for M in (65536, 131072, 32768):
        for K in (16, 64, 128, 256, 512):
            x = torch.randn(M, K, device="cuda", generator=g); W = torch.randn(N, K, device="cuda", generator=g)
            row = ["N %4d M %6d K %4d" % (N, M, K)]
            for env in ("1", "0"):
                os.environ["EGP_GEMM_ROWS"] = env
                us = t(lambda: gemm(x, W, True, True))
                row.append("%s %7.1f us" % ("rows" if env == "1" else "tile", us))
            print(" | ".join(row), flush=True)
