"""T_update of the bench workload, median of n updates on one sampled batch (for A/B runs of library builds)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config
from egopose_amd.physics import default_threads
from egopose_amd.train import Trainer
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp(prefix="egp_ut_"); write_synthetic_dataset(root, "subject_03", device_index=0); os.chdir(root)
cfg = Config("subject_03", create_dirs=False)
tr = Trainer(cfg, dev, torch.float32, num_envs=1024, num_threads=max(2, default_threads()), num_groups=2)
tr.iteration(0, cfg.min_batch_size)
batch, log = tr.agent.sample(cfg.min_batch_size)
ts = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 7):
    torch.cuda.synchronize(); t0 = time.time(); tr.agent.update_params(batch); torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
ts.sort()
print("T_update ms: median %.2f  min %.2f  max %.2f  (%d samples)" % (ts[len(ts) // 2], ts[0], ts[-1], len(batch.masks)))
tr.close()
