"""Device kernels of ONE AgentEgo.update_params call on the bench workload (torch profiler): name, calls, total time."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config
from egopose_amd.physics import default_threads
from egopose_amd.train import Trainer

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp(prefix="egp_uk_"); write_synthetic_dataset(root, "subject_03", device_index=0); os.chdir(root)
cfg = Config("subject_03", create_dirs=False)
tr = Trainer(cfg, dev, torch.float32, num_envs=1024, num_threads=max(2, default_threads()), num_groups=2)
for it in range(2):
    tr.iteration(it, cfg.min_batch_size)
batch, log = tr.agent.sample(cfg.min_batch_size)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    tr.agent.update_params(batch)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total / 1e3) for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"]
if not rows:
    rows = [(e.key, e.count, e.device_time_total / 1e3) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print("device time of one update: %.1f ms in %d kernel names" % (tot, len(rows)))
for k, c, ms in rows[:45]:
    print("%8.2f ms %5d  %s" % (ms, c, k[:130]))
