"""Does the rollout run faster when a few CUs are kept arithmetically busy (clock state)? T_sample with and without a burn kernel
on a side stream for the length of each rollout: python tools/probes/burn_probe.py [blocks]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egopose_amd import _lib as L
from egopose_amd.bench_support import write_synthetic_dataset
from egopose_amd.config import Config
from egopose_amd.physics import default_threads
from egopose_amd.train import Trainer
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp(prefix="egp_bp_"); write_synthetic_dataset(root, "subject_03", device_index=0); os.chdir(root)
cfg = Config("subject_03", create_dirs=False)
tr = Trainer(cfg, dev, torch.float32, num_envs=1024, num_threads=max(2, default_threads()), num_groups=2)
tr.iteration(0, cfg.min_batch_size)
lib = L.load()
sink = torch.zeros(1, device=dev)
side = torch.cuda.Stream()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for rnd in range(3):
    for burn in (0, 1):
        ts = []
        for _ in range(7):
            if burn:
                L.check(lib.egp_debug_burn(95000, blocks, sink.data_ptr(), side.cuda_stream), "burn")
            batch, log = tr.agent.sample(cfg.min_batch_size)
            ts.append(log.sample_time * 1e3)
            torch.cuda.synchronize()
        ts.sort()
        print("round %d burn %d (%d blocks): T_sample median %.2f min %.2f max %.2f" % (rnd, burn, blocks, ts[3], ts[0], ts[-1]))
tr.close()
