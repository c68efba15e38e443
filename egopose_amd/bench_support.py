"""Shared set-up for bench.py, smoke() and the GPU integration tests: a synthetic dataset on disk in the
reference's formats, generated with the product's own kernels (K7 features + backend FK)."""
from __future__ import annotations

import os


def write_synthetic_dataset(root, cfg_id="subject_03", device_index=0, n_takes=8, n_frames=2000, seed=1, cnn_dim=128):
    from .hip import EgpContext
    from .physics import SurrogatePhysics
    from .presets import config_params
    from .skeleton import load_skeleton
    from .synthetic import make_dataset
    sk = load_skeleton()
    p = config_params(cfg_id)
    ctx = EgpContext(sk, p["jkp"], p["jkd"], p["a_ref"], p["a_scale"], p["torque_lim"], p["b_diffw"], p["reward_weights"],
                     episode_len=p["episode_len"], device=device_index)
    phys = SurrogatePhysics(sk, 1)
    try:
        return make_dataset(os.path.abspath(root), ctx, phys, cfg_id, n_takes=n_takes, n_frames=n_frames, cnn_dim=cnn_dim, seed=seed)
    finally:
        phys.close()
        ctx.close()
