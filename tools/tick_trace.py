"""Per-tick timeline of one rollout of the bench workload: stepped envs and the Python thread's wait per env-step.
Usage: python tools/tick_trace.py [envs]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config
from egopose_amd.train import Trainer
from egopose_amd.physics import default_threads
envs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp(prefix="egp_trace_"); write_synthetic_dataset(root, "subject_03", device_index=0); os.chdir(root)
cfg = Config("subject_03", create_dirs=False)
tr = Trainer(cfg, dev, torch.float32, num_envs=envs, num_threads=max(2, default_threads()), num_groups=2)
tr.agent._get_rollout().trace_ticks = True          # (before the first update: it prepares the next rollout's set-up)
for it in range(3):
    log, ts, tu, n = tr.iteration(it, cfg.min_batch_size)
ro = tr.agent._get_rollout()
t = np.array(ro.tick_trace)
print("T_sample %.3f s, %d env-steps of groups, sum wait %.1f ms post %.1f ms reset %.1f ms" % (ts, len(t), t[:, 3].sum() * 1e3, t[:, 4].sum() * 1e3, t[:, 5].sum() * 1e3))
print("tick range | group-steps | mean stepped envs | mean wait us | mean post us | mean reset us")
for lo, hi in ((0, 16), (16, 48), (48, 64), (64, 80), (80, 100), (100, 125), (125, 150), (150, 175), (175, 200), (200, 225), (225, 260)):
    m = (t[:, 1] >= lo) & (t[:, 1] < hi)
    if m.any():
        print("%3d-%3d | %4d | %7.1f | %7.1f | %6.1f | %6.1f" % (lo, hi, m.sum(), t[m, 2].mean(), t[m, 3].mean() * 1e6, t[m, 4].mean() * 1e6, t[m, 5].mean() * 1e6))
