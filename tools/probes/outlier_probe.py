"""Where do the slow rollouts come from? 40 rollouts with the cyclic GC on / off: distribution of T_sample and of the per-tick wait."""
import gc, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config
from egopose_amd.physics import default_threads
from egopose_amd.train import Trainer
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp(prefix="egp_op_"); write_synthetic_dataset(root, "subject_03", device_index=0); os.chdir(root)
cfg = Config("subject_03", create_dirs=False)
tr = Trainer(cfg, dev, torch.float32, num_envs=1024, num_threads=max(2, default_threads()), num_groups=2)
tr.iteration(0, cfg.min_batch_size)
for mode in ("gc on", "gc off", "gc on", "gc off"):
    if mode == "gc off":
        gc.collect(); gc.disable()
    else:
        gc.enable()
    ts = []
    for _ in range(30):
        batch, log = tr.agent.sample(cfg.min_batch_size)
        ts.append(log.sample_time * 1e3)
    ts = np.array(ts)
    print("%-6s: median %.1f  p90 %.1f  max %.1f  mean %.1f   slow (> 1.1 x median): %d of %d   gc counts %s" % (
        mode, np.median(ts), np.percentile(ts, 90), ts.max(), ts.mean(), int((ts > 1.1 * np.median(ts)).sum()), len(ts), gc.get_count()))
gc.enable()
tr.close()
