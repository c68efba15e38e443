"""Can an LSTM sweep and an MLP GEMM of the update share the GPU? Each alone, then both enqueued on two streams.
(The question behind running the value net's and the policy net's update on two streams.)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egopose_amd import _lib as L, gemm as G
T, B, H = 220, 1280, 64
lib = L.load()
dev = torch.device("cuda", 0)
gx = torch.randn(T, B, 4 * H, device=dev) * 0.5
w = torch.randn(4 * H, H, device=dev) * 0.1
h = torch.empty(T, B, H, device=dev); gates = torch.empty(T, B, 4 * H, device=dev); cells = torch.empty(T, B, H, device=dev)
dh = torch.randn(T, B, H, device=dev); dpre = torch.empty(T, B, 4 * H, device=dev)
x = torch.randn(131072, 300, device=dev); W = torch.randn(200, 300, device=dev) * 0.05; b = torch.zeros(200, device=dev)
out = torch.empty(131072, 200, device=dev)
dy = torch.randn(131072, 200, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def lstm(stream):
    with torch.cuda.stream(stream):
        L.check(lib.egp_lstm_fwd_f32(p(gx), p(w), T, B, H, 0, p(h), p(gates), p(cells), L.current_stream()), "fwd")


def lstm_b(stream):
    with torch.cuda.stream(stream):
        L.check(lib.egp_lstm_bwd_f32(p(dh), p(gates), p(cells), p(w), T, B, H, 0, p(dpre), L.current_stream()), "bwd")


def gemm(stream, n=4):
    with torch.cuda.stream(stream):
        for _ in range(n):
            G.gemm(x, W, True, True, bias=b, relu=True, out=out)


def wgrad(stream, n=2):
    with torch.cuda.stream(stream):
        for _ in range(n):
            G.linear_wgrad(dy, x, want_bias=True)


def t(fs, reps=10):
    for _ in range(2):
        for f in fs: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_event(e0); s2.wait_event(e0)
    for _ in range(reps):
        for f in fs: f()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, a, bb in (("lstm fwd | 4 fwd gemms", lambda: lstm(s1), lambda: gemm(s2)), ("lstm bwd | 4 fwd gemms", lambda: lstm_b(s1), lambda: gemm(s2)),
                    ("lstm fwd | 2 wgrads", lambda: lstm(s1), lambda: wgrad(s2)), ("4 fwd gemms | 2 wgrads", lambda: gemm(s1), lambda: wgrad(s2))):
    ta, tb, tab = t([a]), t([bb]), t([a, bb])
    print("%-26s alone %.0f + %.0f = %.0f us, together %.0f us (%.0f %% of the sum)" % (name, ta, tb, ta + tb, tab, 100 * tab / (ta + tb)))
