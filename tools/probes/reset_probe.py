"""Where the rollout's `reset` time goes: the pieces of LockstepRollout._reset_slots for all slots and for a handful."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config
from egopose_amd.physics import default_threads
from egopose_amd.train import Trainer
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp(prefix="egp_rp_"); write_synthetic_dataset(root, "subject_03", device_index=0); os.chdir(root)
cfg = Config("subject_03", create_dirs=False)
tr = Trainer(cfg, dev, torch.float32, num_envs=1024, num_threads=max(2, default_threads()), num_groups=2)
tr.iteration(0, cfg.min_batch_size)
ro = tr.agent._get_rollout()
ex = ro.experts


def T(f, n=5):
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.time(); f(); torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e6)
    return sorted(ts)[len(ts) // 2]


for cnt in (1024, 8):
    ids = np.arange(cnt)
    ro._pool = None
    print("== %d slots" % cnt)
    print("  draw_episodes (pool refill: gather + LSTM)  %8.0f us" % T(lambda: (setattr(ro, "_pool", None), ro._draw_episodes(cnt))))
    print("  draw_episodes (from the pool)               %8.0f us" % T(lambda: ro._draw_episodes(cnt)))
    e_ind, s_ind, ctx_rows = ro._draw_episodes(cnt)
    rows = ex.take_offset[e_ind] + s_ind
    print("  expert rows (numpy gather + copy)           %8.0f us" % T(lambda: (ex.qpos[rows].copy(), ex.qvel[rows].copy())))
    qpos, qvel = ex.qpos[rows].copy(), ex.qvel[rows].copy()
    print("  engine.reset                                %8.0f us" % T(lambda: ro.engine.reset(ids, qpos, qvel)))
    ids_d = ro.up(ids)
    print("  up(ids)                                     %8.0f us" % T(lambda: ro.up(ids)))
    print("  v_out[ids] = ctx rows                       %8.0f us" % T(lambda: ro.v_out.__setitem__(ids_d, ctx_rows)))
    print("  whole _reset_slots                          %8.0f us" % T(lambda: ro._reset_slots(ids)))
tr.close()
