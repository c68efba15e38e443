"""One GEMM shape, repeated (for rocprofv3 --pmc passes). Usage: gemm_one.py [fwd|dgrad|wgrad] [terms] [rows] [k] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egopose_amd.gemm import linear_fwd, linear_dgrad, linear_wgrad
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
terms = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = int(sys.argv[3]) if len(sys.argv) > 3 else 139264
k = int(sys.argv[4]) if len(sys.argv) > 4 else 243
m = int(sys.argv[5]) if len(sys.argv) > 5 else 300
x = torch.randn(n, k, device="cuda"); w = torch.randn(m, k, device="cuda") * 0.1; b = torch.randn(m, device="cuda")
gy = torch.randn(n, m, device="cuda"); h = torch.randn(n, k, device="cuda")
f = {"fwd": lambda: linear_fwd(x, w, b, True, terms=terms), "dgrad": lambda: linear_dgrad(gy, w, mask=h, terms=terms),
     "wgrad": lambda: linear_wgrad(gy, x, terms=terms)}[kind]
for _ in range(10):
    f()
torch.cuda.synchronize()
