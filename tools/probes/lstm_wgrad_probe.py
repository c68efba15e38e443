"""Time of the LSTM weight-gradient products at the bench's shapes (T*B window rows x P*4H gate columns)."""
import sys, torch
from egopose_amd import gemm as G
R = int(sys.argv[1]) if len(sys.argv) > 1 else 154000
dev = torch.device("cuda")
dpre = torch.randn(R, 1024, device=dev); x = torch.randn(R, 128, device=dev); h = torch.randn(R, 128, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
print("rows", R)
print("dW_ih  (1024 x 128 over R rows)  %.0f us" % t(lambda: G.linear_wgrad(dpre, x, want_bias=False)))
print("dW_hh  (256 x 64, strided views) %.0f us each" % t(lambda: G.linear_wgrad(dpre[:, 256:512], h[:, 64:128], want_bias=False)))
print("d_x-free: read of dpre alone (sum over rows) %.0f us" % t(lambda: dpre.sum(0)))
F = 16000
tab = torch.randn(F, 128, device=dev); D = torch.randn(F, 1024, device=dev)
print("dW_ih over %d frames %.0f us" % (F, t(lambda: G.linear_wgrad(D, tab, want_bias=False))))
idx = torch.randint(0, F, (R,), device=dev)
print("index_add of dpre into frames (atomics) %.0f us" % t(lambda: torch.zeros(F, 1024, device=dev).index_add_(0, idx, dpre)))
