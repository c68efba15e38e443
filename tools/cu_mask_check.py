#!/usr/bin/env python3
"""What the resident K1 does when fewer CUs are to be had: prints the occupancy calculator's and the probe's capacity and the engine
form taken for `--envs` slots. Run under a CU mask, e.g. HSA_CU_MASK=0:0-239 python tools/cu_mask_check.py"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--step", action="store_true", help="also run two env-steps from a fixed state and print a checksum of the final qpos")
    a = ap.parse_args()
    import torch
    from egopose_amd.hip import EgpContext
    from egopose_amd.physics import RolloutEngine, SurrogatePhysics
    from egopose_amd.presets import subject_03_params
    from egopose_amd.skeleton import load_skeleton
    sk = load_skeleton()
    p = subject_03_params()
    ctx = EgpContext(sk, p["jkp"], p["jkd"], p["a_ref"], p["a_scale"], p["torque_lim"], p["b_diffw"], p["reward_weights"], device=0)
    ph = SurrogatePhysics(sk, a.envs)
    eng = RolloutEngine(ctx, ph, a.envs, n_threads=4, n_groups=a.groups)
    out = {"envs": a.envs, "cus_reported": torch.cuda.get_device_properties(0).multi_processor_count,
           "envs_per_wave": eng.envs_per_wave, "resident_capacity": eng.resident_capacity, "substeps_per_launch": eng.substeps_per_launch,
           "HSA_CU_MASK": os.environ.get("HSA_CU_MASK"), "ROC_GLOBAL_CU_MASK": os.environ.get("ROC_GLOBAL_CU_MASK")}
    if a.step:
        import numpy as np
        rng = np.random.RandomState(0)
        q0 = np.tile(np.r_[0, 0, 1.0, 1, 0, 0, 0, np.zeros(52)], (a.envs, 1))
        q0[:, 7:] += rng.normal(size=(a.envs, 52)) * 0.1
        eng.reset(np.arange(a.envs), q0, rng.normal(size=(a.envs, 58)) * 0.1)
        act = torch.as_tensor(rng.normal(size=(a.envs, 52)) * 0.1, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):
            for g in range(a.groups):
                eng.step_async(g, act)
            for g in range(a.groups):
                eng.wait(g)
        torch.cuda.synchronize()
        q = eng.qpos.cpu().numpy()
        out["qpos_sum"] = float(q.sum())
        out["qpos_abs_sum"] = float(np.abs(q).sum())
        out["qpos_probe"] = [float(x) for x in q[[0, a.envs // 2, a.envs - 1], 10]]
    eng.close(); ph.close(); ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
