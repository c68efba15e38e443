import os, sys, tempfile, torch, numpy as np
sys.path.insert(0, os.getcwd())
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config
from egopose_amd.train import Trainer
root = tempfile.mkdtemp(prefix="egp_sp_"); write_synthetic_dataset(root, "subject_03", device_index=0, n_takes=2, n_frames=400); os.chdir(root)
def run(share):
    os.environ["EGP_SHARE_TRAIN_CONTEXT"] = share
    cfg = Config("subject_03", create_dirs=False); cfg.env_episode_len = 12; cfg.num_optim_epoch = 2
    tr = Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=32, num_threads=4, num_groups=2)
    tr.iteration(0, 512)
    ps = {k: v.detach().clone() for n in ("value_net", "value_vs_net", "policy_vs_net", "policy_net") for k, v in getattr(tr, n).state_dict(prefix=n + ".").items()}
    tr.close()
    return ps
a, b, c = run("0"), run("0"), run("1")
for name, x, y in (("0 vs 0", a, b), ("0 vs 1", a, c)):
    worst = max(((x[k] - y[k]).abs().max().item(), k) for k in x)
    print(name, "max |d| = %.3e at %s" % worst)
