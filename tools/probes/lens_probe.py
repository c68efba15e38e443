import os, sys, tempfile
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/egopose_amd") else os.getcwd())
import numpy as np, torch
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config
from egopose_amd.physics import default_threads
from egopose_amd.train import Trainer
dev = torch.device("cuda", 0)
root = tempfile.mkdtemp(prefix="egp_l_"); write_synthetic_dataset(root, "subject_03", device_index=0); os.chdir(root)
cfg = Config("subject_03", create_dirs=False)
tr = Trainer(cfg, dev, torch.float32, num_envs=1024, num_threads=max(2, default_threads()), num_groups=2)
for it in range(2):
    batch, log = tr.agent.sample(cfg.min_batch_size)
    m = np.asarray(batch.masks)
    ends = np.nonzero(m == 0)[0]; starts = np.r_[0, ends[:-1] + 1]; lens = ends - starts + 1
    print("episodes %d, steps %d, len mean %.1f, hist %s" % (len(lens), len(m), lens.mean(), np.histogram(lens, bins=[0, 25, 50, 100, 150, 199, 201])[0]))
    q = np.sort(lens)[::-1]
    pad = len(q) % 4
    qm = q[:len(q) - pad].reshape(-1, 4).max(1) if len(q) >= 4 else q
    r = lens[:len(lens) - pad].reshape(-1, 4).max(1)
    print("  sum(len+20) = %d of T*B = %d; per-quad max: sorted %.1f, unsorted %.1f (of %d)" % ((lens + 20).sum(), (lens.max() + 20) * len(lens), qm.mean() + 20, r.mean() + 20, lens.max() + 20))
    tr.agent.update_params(batch)
