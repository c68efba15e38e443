import torch, sys
sys.path.insert(0, '.')
from egopose_amd import gemm as G
dev = torch.device('cuda'); R = 134000
dy = torch.randn(R, 1, device=dev); h = torch.randn(R, 200, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for sp in (128, 256, 512, 1024):
    G.pick_splits = lambda M, n_out, K, sp=sp: sp
    print(sp, "splits: %.1f us" % t(lambda: G.linear_wgrad(dy, h, want_bias=True)))
