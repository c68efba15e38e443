"""Is the bias+ReLU epilogue of hipBLASLt reachable from torch here, and is it faster than addmm + relu?"""
import torch, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from egopose_amd import gemm_tuning
dev = "cuda"; n = 139264
def bench(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for tuned in (False, True):
    if tuned: print("tuned picks:", gemm_tuning.enable())
    for k, m in ((243, 300), (300, 200)):
        x = torch.randn(n, k, device=dev); w = torch.randn(m, k, device=dev); b = torch.randn(m, device=dev)
        a = bench(lambda: torch.relu(torch.addmm(b, x, w.t())))
        c = bench(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False))
        same = torch.allclose(torch.relu(torch.addmm(b, x, w.t())), torch._addmm_activation(b, x, w.t(), use_gelu=False), atol=1e-3, rtol=1e-4)
        print("%dx%d  addmm+relu %.1f us   _addmm_activation %.1f us   same=%s" % (k, m, a, c, same))
