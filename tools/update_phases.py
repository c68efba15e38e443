"""Where T_update goes: wall time (with device syncs) of the phases of AgentEgo.update_params on the bench workload."""
import os, sys, tempfile, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config
from egopose_amd.train import Trainer
from egopose_amd.physics import default_threads
from egopose_amd import agent as A, nets as N
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp(prefix="egp_upd_"); write_synthetic_dataset(root, "subject_03", device_index=0); os.chdir(root)
cfg = Config("subject_03", create_dirs=False)
tr = Trainer(cfg, dev, torch.float32, num_envs=1024, num_threads=max(2, default_threads()), num_groups=2)
acc = collections.OrderedDict()


def timed(obj, name, label):
    fn = getattr(obj, name)

    def wrap(*a, **k):
        torch.cuda.synchronize(); t = time.time()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + time.time() - t
        return r
    setattr(obj, name, wrap)


ag = tr.agent
timed(ag, "_load_batch", "load_batch")
timed(ag.policy_vs_net, "initialize", "vs_net.initialize x2")
timed(ag.value_vs_net, "initialize", "vs_net.initialize x2")
timed(ag, "_advantages", "advantages (GAE)")
timed(ag, "update_policy", "update_policy (log-probs + 10 epochs)")
timed(ag, "ppo_loss", "  of which ppo_loss forward x10")
timed(ag, "_sync_grads", "  grad sync")
timed(ag.optimizer_policy, "step", "  optimizer steps")
timed(ag.optimizer_value, "step", "  optimizer steps")
for it in range(4):
    acc.clear()
    m0 = torch.cuda.memory_reserved()
    log, ts, tu, n = tr.iteration(it, cfg.min_batch_size)
    print("iteration %d: T_update %.1f ms (%d steps), allocator reserved %.2f -> %.2f GB" % (it, tu * 1e3, n, m0 / 2**30, torch.cuda.memory_reserved() / 2**30))
    if it in (1, 3):
        for k, v in acc.items():
            print("    %-42s %7.1f ms" % (k, v * 1e3))
