"""Time the persistent LSTM kernels alone (HIP events): python tools/lstm_probe.py [T B H]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egopose_amd import _lib as L
T, B, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (220, 1280, 64)
lib = L.load()
dev = torch.device("cuda", 0)
gx = torch.randn(T, B, 4 * H, device=dev) * 0.5
w = torch.randn(4 * H, H, device=dev) * 0.1
h = torch.empty(T, B, H, device=dev); gates = torch.empty(T, B, 4 * H, device=dev); cells = torch.empty(T, B, H, device=dev)
dh = torch.randn(T, B, H, device=dev); dpre = torch.empty(T, B, 4 * H, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
s = L.current_stream()


def bench(f, reps=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


fwd = lambda: L.check(lib.egp_lstm_fwd_f32(p(gx.clone()), p(w), T, B, H, 0, p(h), p(gates), p(cells), s), "fwd")
cl = lambda: gx.clone()
bwd = lambda: L.check(lib.egp_lstm_bwd_f32(p(dh), p(gates), p(cells), p(w), T, B, H, 0, p(dpre), s), "bwd")
t_clone = bench(cl)
print("T %d B %d H %d: fwd %.1f us  bwd %.1f us  (per step %.2f / %.2f us)" % (
    T, B, H, bench(fwd) - t_clone, bench(bwd), (bench(fwd) - t_clone) / T, bench(bwd) / T))
