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
    eng.close(); ph.close(); ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
