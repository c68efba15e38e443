"""Regenerate egopose_amd/assets/tunableop/gfx950.csv on an MI355X: run PPO iterations of the bench workload with
TunableOp tuning on, over min-batch sizes that reach the neighbouring row / episode buckets.
Usage: python tools/tune_update.py OUT.csv [task] [envs]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egopose_amd import gemm_tuning
out = os.path.abspath(sys.argv[1])
task = sys.argv[2] if len(sys.argv) > 2 else "egomimic"
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config, ForecastConfig
from egopose_amd.train import Trainer
from egopose_amd.physics import default_threads
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
root = tempfile.mkdtemp(prefix="egp_tune_")
write_synthetic_dataset(root, "subject_03", device_index=0)
os.chdir(root)
cfg = (ForecastConfig if task == "egoforecast" else Config)("subject_03", create_dirs=False)
tr = Trainer(cfg, dev, torch.float32, num_envs=envs, num_threads=max(2, default_threads()), num_groups=2)
assert gemm_tuning.enable(tune=True, out_file=out)
seen = set()
for it, mb in enumerate([50000, 44000, 47000, 53000, 56000, 41000, 59000, 50000]):
    t0 = time.time()
    log, ts, tu, n = tr.iteration(it, mb)
    print("iter %d min_batch %d steps %d  T_update %.2f s (wall %.1f)" % (it, mb, n, tu, time.time() - t0), flush=True)
import torch.cuda.tunable as tn
res = tn.get_results()
print(len(res), "tuned entries ->", out)
