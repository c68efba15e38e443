"""T_sample of the bench workload (median over n rollouts) under the current environment switches -- for in-lease A/B runs."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("EGP_PROBE_SPIN_SYNC") == "1":        # hipDeviceScheduleSpin before the runtime creates its context
    import ctypes
    _hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(spin) ->", _hip.hipSetDeviceFlags(1))
import torch
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config
from egopose_amd.physics import default_threads
from egopose_amd.train import Trainer
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp(prefix="egp_st_"); write_synthetic_dataset(root, "subject_03", device_index=0); os.chdir(root)
cfg = Config("subject_03", create_dirs=False)
tr = Trainer(cfg, dev, torch.float32, num_envs=1024, num_threads=int(os.environ.get("EGP_PROBE_THREADS", "0")) or max(2, default_threads()), num_groups=int(os.environ.get("EGP_PROBE_GROUPS", "2")))
tr.iteration(0, cfg.min_batch_size)
ts, ns = [], []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 7):
    batch, log = tr.agent.sample(cfg.min_batch_size)
    ts.append(log.sample_time * 1e3); ns.append(len(batch.masks))
ts.sort()
tm = tr.agent._get_rollout().timing
print("T_sample ms: median %.2f  min %.2f  max %.2f  (%d steps, %d ticks; wait %.1f policy %.1f post %.1f reset %.1f ms; small ticks %d / %.1f ms)" % (
    ts[len(ts) // 2], ts[0], ts[-1], ns[-1], tm["ticks"], tm["wait"] * 1e3, tm["policy"] * 1e3, tm["post"] * 1e3, tm["reset"] * 1e3,
    tm["small_group_ticks"], tm["small_group_tick_s"] * 1e3))
print("   outside the tick loop: set-up %.1f ms, batch assembly %.1f ms" % (tm.get("setup", 0) * 1e3, tm.get("assemble", 0) * 1e3))
print("   engine: phys %.1f ms  gpu_wait %.1f ms (timekeeper threads, summed over groups), k1 %.1f ms / %d launches" % (
    tm.get("phys_s", 0) * 1e3, tm.get("gpu_wait_s", 0) * 1e3, tm.get("k1_ms", 0), tm.get("k1_launches", 0)))
tr.close()
