"""One product repeated (for rocprofv3 passes): python tools/gemm_one.py [rows|tile] [case] -- case: fwd1 (134656 x 243 -> 300, bias + ReLU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["EGP_GEMM_ROWS"] = "0" if (len(sys.argv) > 1 and sys.argv[1] == "tile") else "1"
import torch
from egopose_amd.gemm import gemm
M = 134656
g = torch.Generator(device="cuda").manual_seed(0)
x, W, b = torch.randn(M, 243, device="cuda", generator=g), torch.randn(300, 243, device="cuda", generator=g), torch.randn(300, device="cuda", generator=g)
for _ in range(12):
    gemm(x, W, True, True, bias=b, relu=True)
torch.cuda.synchronize()
