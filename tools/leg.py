#!/usr/bin/env python3
"""One of bench.py's legs alone (same `run_leg`, same workload): `python tools/leg.py device [--envs N --groups G --steps K]`.
   names: headline | device (changing inertia, K8 on the device) | hostfed (changing inertia from the host) | cost20 (20 us substep)"""
import argparse
import json
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

LEGS = {"headline": {}, "device": {"EGP_SURROGATE_ALWAYS_DIRTY": "1", "EGP_DEVICE_DYNAMICS": "1"},
        "hostfed": {"EGP_SURROGATE_ALWAYS_DIRTY": "1"}, "cost20": {"EGP_SURROGATE_SUBSTEP_US": "20"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("legs", nargs="+", choices=sorted(LEGS))
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--env", action="append", default=[], help="extra KEY=VALUE for the leg's environment")
    a = ap.parse_args()
    import torch
    from egopose_amd.bench_support import write_synthetic_dataset
    from egopose_amd.config import Config
    from egopose_amd.physics import default_threads
    from egopose_amd.train import Trainer
    root = tempfile.mkdtemp(prefix="egp_leg_")
    write_synthetic_dataset(root, "subject_03", device_index=0)
    os.chdir(root)
    dev = torch.device("cuda", 0)
    n_threads = a.threads or max(a.groups, default_threads(share=1, device_index=0))
    cfg = Config("subject_03", create_dirs=False)
    mk = lambda: Trainer(Config("subject_03", create_dirs=False), dev, torch.float32, num_envs=a.envs, num_threads=n_threads, num_groups=a.groups)
    extra = dict(kv.split("=", 1) for kv in a.env)
    for name in a.legs:
        r = bench.run_leg(mk, a.steps, a.warmup, cfg.min_batch_size * a.envs // 1024, 8, dict(LEGS[name], **extra))
        r["leg"] = name
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)


if __name__ == "__main__":
    main()
