"""Is the grouped forward LSTM sweep bound by its bytes? Same launch geometry (1 280 workgroups of 4 sequences, T = 220), with and
without the save-set of the backward pass (gates 1 kB + cells 256 B per sequence-step): kernel time by HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egopose_amd.nets import RNN
torch.manual_seed(0)
T, B = 220, 2560
x = torch.randn(T, B, 128, device="cuda")
rnn = RNN(128, 128, "lstm", bi_dir=True).cuda()


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def fwd_train():
    return rnn(x)


def fwd_eval():
    with torch.no_grad():
        return rnn(x)


print("forward incl. projection GEMM, training (gates + cells saved): %.3f ms" % timed(fwd_train))
print("forward incl. projection GEMM, no save-set:                    %.3f ms" % timed(fwd_eval))
from torch.profiler import profile, ProfilerActivity
for name, fn in (("train", fwd_train), ("eval", fwd_eval)):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    for e in prof.key_averages():
        if "k_lstm" in e.key:
            print("  %s: %s  %.1f us per launch" % (name, e.key[:60], e.device_time_total / e.count))
