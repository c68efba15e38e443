"""cProfile of the host side of rollouts (where the Python thread's time outside the waits goes)."""
import cProfile, pstats, io, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config
from egopose_amd.physics import default_threads
from egopose_amd.train import Trainer
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp(prefix="egp_sp_"); write_synthetic_dataset(root, "subject_03", device_index=0); os.chdir(root)
cfg = Config("subject_03", create_dirs=False)
tr = Trainer(cfg, dev, torch.float32, num_envs=1024, num_threads=max(2, default_threads()), num_groups=2)
tr.iteration(0, cfg.min_batch_size)
for _ in range(2):
    tr.agent.sample(cfg.min_batch_size)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.agent.sample(cfg.min_batch_size)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
tr.close()
