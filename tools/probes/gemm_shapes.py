"""Which products the update spends its GEMM time on: per (form, M, N, K) calls and device time per PPO iteration
(HIP events around every egopose_amd.gemm.gemm call of two iterations of the bench workload)."""
import collections, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egopose_amd import gemm as G
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config
from egopose_amd.physics import default_threads
from egopose_amd.train import Trainer

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp(prefix="egp_gs_"); write_synthetic_dataset(root, "subject_03", device_index=0); os.chdir(root)
cfg = Config("subject_03", create_dirs=False)
tr = Trainer(cfg, dev, torch.float32, num_envs=1024, num_threads=max(2, default_threads()), num_groups=2)
events = []
inner = G.gemm


def timed(A, B, a_kcontig=True, b_kcontig=True, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = inner(A, B, a_kcontig, b_kcontig, **kw)
    e1.record()
    M = kw["a_rows"].shape[0] if kw.get("a_rows") is not None else (A.shape[0] if a_kcontig else A.shape[1])
    K = (A.shape[1] if a_kcontig else A.shape[0]) + (kw["a2"].shape[1] if kw.get("a2") is not None else 0)
    if kw.get("b_krows") is not None:
        K = kw["b_krows"].shape[0]
    N = (B.shape[0] if b_kcontig else B.shape[1]) + (kw["b2"].shape[1] if kw.get("b2") is not None else 0)
    tag = "%s%s%s%s" % ("kc" if a_kcontig else "rc", "kc" if b_kcontig else "rc", "+gather" if any(kw.get(k) is not None for k in ("a_rows", "b_krows", "c_rows")) else "",
                        "+mask" if kw.get("mask") is not None else "")
    events.append(((tag, M, N, K, int(kw.get("splits", 1))), e0, e1))
    return out


G.gemm = timed
for it in range(3):
    events.clear()
    log, ts, tu, n = tr.iteration(it, cfg.min_batch_size)
torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in events:
    a = acc[key]; a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in acc.values())
print("T_update %.1f ms, T_sample %.1f ms; %d gemm calls, %.1f ms inside them (events include launch gaps)" % (tu * 1e3, ts * 1e3, len(events), tot))
for key, (c, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    fl = 2.0 * key[1] * key[2] * key[3]
    print("  %-14s M %7d N %5d K %7d splits %3d : %3d calls %7.2f ms  (%6.1f us each, %5.1f TF/s)" % (key[0], key[1], key[2], key[3], key[4], c, ms, ms / c * 1e3, fl * c / ms / 1e9))
