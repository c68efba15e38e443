"""One PPO epoch of AgentEgo.update_params on the bench workload, call by call: every egp_gemm_f32 product and every grouped LSTM
sweep bracketed by HIP events on the stream it runs on, with its shape, operand forms, algorithmic bytes and float32-equivalent rate.
    python tools/epoch_trace.py [--epoch 5] > gpurun_out/epoch_trace.txt
(The events add ~1-2 us of stream work per call; the sums are therefore a little above the profiler's kernel sums.)"""
import argparse, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config
from egopose_amd.physics import default_threads
from egopose_amd.train import Trainer
from egopose_amd import gemm as G, _lib as L, optim as O

ap = argparse.ArgumentParser()
ap.add_argument("--epoch", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp(prefix="egp_et_"); write_synthetic_dataset(root, "subject_03", device_index=0); os.chdir(root)
cfg = Config("subject_03", create_dirs=False)
tr = Trainer(cfg, dev, torch.float32, num_envs=1024, num_threads=max(2, default_threads()), num_groups=2)
for it in range(2):
    tr.iteration(it, cfg.min_batch_size)
batch, log = tr.agent.sample(cfg.min_batch_size)
torch.cuda.synchronize()

calls = []          # (epoch, label, start event, end event, bytes, flops)
state = {"epoch": -1}


def bracket(label, nbytes, flops, fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = fn()
    b.record()
    calls.append((state["epoch"], label, a, b, nbytes, flops))
    return r


_gemm = G.gemm


def gemm(A, B, a_kcontig=True, b_kcontig=True, **kw):
    M, K = (A.shape if a_kcontig else A.shape[::-1])
    N, Kb = (B.shape if b_kcontig else B.shape[::-1])
    if kw.get("a_rows") is not None:
        M = kw["a_rows"].shape[0]
    if kw.get("a2") is not None:
        K = K + kw["a2"].shape[1]
    if kw.get("a_krows") is not None:
        K = kw["a_krows"].shape[0]
    if kw.get("b_krows") is not None:
        Kb = kw["b_krows"].shape[0]
    if kw.get("b2") is not None:
        N = N + kw["b2"].shape[1]
    form = "%s%s" % ("A[m][k]" if a_kcontig else "A[k][m]", " B[n][k]" if b_kcontig else " B[k][n]")
    extra = ",".join(k for k in ("bias", "mask", "a_rows", "a2", "b_krows", "b2", "c_rows", "a_krows", "want_bias_grad") if kw.get(k) is not None and kw.get(k) is not False)
    if kw.get("relu"):
        extra += ",relu"
    label = "gemm %7d x %4d x %7d  %s  splits=%d  %s" % (M, N, K, form, kw.get("splits", 1), extra)
    nbytes = 4 * (M * K + N * K + M * N + (M * N if kw.get("mask") is not None else 0))
    return bracket(label, nbytes, 2.0 * M * N * K, lambda: _gemm(A, B, a_kcontig, b_kcontig, **kw))


G.gemm = gemm
lib = L.load()


class LibProxy:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in ("egp_lstm_group_fwd_len_f32", "egp_lstm_group_bwd_len_f32"):
            def wrap(*a):
                if name.endswith("fwd_len_f32"):
                    T, B, H, P = a[2], a[3], a[4], a[5]
                else:
                    T, B, H, P = a[5], a[6], a[7], a[8]
                return bracket("%s T=%d B=%d H=%d P=%d" % (name[4:], T, B, H, P), 0, 2.0 * T * B * P * 4 * H * H, lambda: fn(*a))
            return wrap
        return fn


proxy = LibProxy(lib)
import egopose_amd.lstm as LS
LS.L = type("Lmod", (), {"load": staticmethod(lambda: proxy), "check": staticmethod(L.check), "current_stream": staticmethod(L.current_stream)})
_losses = O.ppo_losses


def losses(*a, **k):
    state["epoch"] += 1
    return bracket("ppo_losses", 0, 0, lambda: _losses(*a, **k))


O.ppo_losses = losses
import egopose_amd.agent as AG
AG.O = O
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
tr.agent.update_params(batch)
e1.record()
torch.cuda.synchronize()
print("update_params: %.2f ms between events, %d bracketed calls" % (e0.elapsed_time(e1), len(calls)))
# an epoch = from the loss launch of epoch e (backward + step follow) to the loss launch of epoch e + 1 (its forward precedes it):
# print the calls recorded with state == epoch (backward of `epoch`, optimizer, forward of `epoch + 1`)
tot = {}
for ep, label, a, b, nb, fl in calls:
    ms = a.elapsed_time(b)
    tot.setdefault(ep, 0.0)
    tot[ep] += ms
    if ep == args.epoch:
        rate = "" if not nb else "  %6.2f TB/s" % (nb / ms / 1e9)
        tf = "" if not fl else "  %6.1f TF/s" % (fl / ms / 1e9)
        print("%8.1f us  %s%s%s" % (ms * 1e3, label, rate, tf))
print("bracketed time per epoch window (ms):", {k: round(v, 2) for k, v in sorted(tot.items())})
