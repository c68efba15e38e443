"""Synthetic MoCap-like dataset in the reference's on-disk formats.

No dataset ships with the reference (``datasets/`` is git-ignored there), so benchmarks and GPU tests
run on generated data: smooth random humanoid motion -> expert features (K7 + FK) and N(0,1) CNN features,
written exactly as the reference's tools would write them:
  datasets/meta/<meta_id>.yml                  keys train/test (+ video_mocap_sync, capture)
  datasets/features/expert_<id>.p              pickle dict[take] -> dict  (gen_expert.py:28-83,99-100)
  datasets/features/cnn_feat_<id>.p            pickle (dict[take] -> (T,128) float64, meta)  (gen_cnn_feature.py:68-70)
  config/egomimic/<id>.yml                     copied from the packaged config
so the unmodified driver flow  Config -> HumanoidEnv.load_experts  is exercised end to end.
"""
from __future__ import annotations

import os
import pickle
import shutil

import numpy as np
import yaml

from .config import _ASSET_CFG, packaged_config


def _q_about(angle, axis):
    axis = np.asarray(axis, float)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([[np.cos(angle / 2)], np.sin(angle / 2) * axis])


def _q_mul(a, b):
    w1, x1, y1, z1 = a
    w0, x0, y0, z0 = b
    return np.array([w1 * w0 - x1 * x0 - y1 * y0 - z1 * z0, w1 * x0 + x1 * w0 + y1 * z0 - z1 * y0,
                     w1 * y0 - x1 * z0 + y1 * w0 + z1 * x0, w1 * z0 + x1 * y0 - y1 * x0 + z1 * w0])


def synth_qpos_sequence(skel, rng, n_frames, fps=30.0):
    """Smooth walking-ish motion: root drifts forward with slow yaw, joints follow a few sinusoids
    inside their MJCF ranges, hands at rest."""
    t = np.arange(n_frames) / fps
    nj = skel.nq - 7
    qpos = np.zeros((n_frames, skel.nq))
    yaw = rng.uniform(-np.pi, np.pi) + 0.25 * np.sin(0.2 * t + rng.uniform(0, 6.28)) + 0.05 * t
    speed = 0.6 + 0.3 * np.sin(0.13 * t + rng.uniform(0, 6.28))
    qpos[:, 0] = np.cumsum(speed * np.cos(yaw)) / fps
    qpos[:, 1] = np.cumsum(speed * np.sin(yaw)) / fps
    qpos[:, 2] = 0.90 + 0.02 * np.sin(2 * np.pi * 1.6 * t + rng.uniform(0, 6.28))
    tilt_axis = rng.normal(size=3)
    tilt = 0.05 * np.sin(2 * np.pi * 0.8 * t + rng.uniform(0, 6.28))
    for i in range(n_frames):
        qpos[i, 3:7] = _q_mul(_q_about(yaw[i], [0, 0, 1]), _q_about(tilt[i], tilt_axis))
    lo, hi = skel.joint_range[:, 0], skel.joint_range[:, 1]
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    center = np.clip(rng.normal(size=nj) * 0.15, -0.5, 0.5) * half * 0.5 + np.clip(mid, -0.4, 0.4)
    amp = np.minimum(half * 0.35, 0.35) * rng.uniform(0.3, 1.0, size=nj)
    joints = center[None, :].repeat(n_frames, 0)
    for _ in range(3):
        f = rng.uniform(0.2, 1.8, size=nj)
        ph = rng.uniform(0, 2 * np.pi, size=nj)
        joints += (amp / 3.0) * np.sin(2 * np.pi * f[None, :] * t[:, None] + ph[None, :])
    qpos[:, 7:] = np.clip(joints, lo, hi)
    addr = skel.body_qposaddr()
    for hand in ("LeftHand", "RightHand"):
        qpos[:, slice(*addr[hand])] = 0.0
    return qpos


def make_dataset(root, ctx, physics, cfg_id="subject_03", n_takes=8, n_frames=2000, cnn_dim=128, seed=1,
                 n_test_takes=1):
    """Write a synthetic dataset + config under ``root``; returns the list of training take names."""
    from .expert import build_expert_take

    cfg = packaged_config(cfg_id)
    os.makedirs(os.path.join(root, "config", "egomimic"), exist_ok=True)
    shutil.copy(os.path.join(_ASSET_CFG, "%s.yml" % cfg_id), os.path.join(root, "config", "egomimic", "%s.yml" % cfg_id))
    os.makedirs(os.path.join(root, "datasets", "meta"), exist_ok=True)
    os.makedirs(os.path.join(root, "datasets", "features"), exist_ok=True)
    rng = np.random.RandomState(seed)
    train = ["synth_%02d" % i for i in range(n_takes)]
    test = ["synth_t%02d" % i for i in range(n_test_takes)]
    experts, feats = {}, {}
    for name in train + test:
        qpos = synth_qpos_sequence(ctx.skel, rng, n_frames)
        experts[name] = build_expert_take(ctx, physics, qpos)
        feats[name] = rng.normal(size=(n_frames, cnn_dim))
    meta = {"train": train, "test": test, "capture": {"fps": 30},
            "video_mocap_sync": {name: [0, 0, n_frames] for name in train + test}}
    with open(os.path.join(root, "datasets", "meta", "%s.yml" % cfg["meta_id"]), "w") as f:
        yaml.safe_dump(meta, f)
    with open(os.path.join(root, "datasets", "features", "expert_%s.p" % cfg["expert_feat"]), "wb") as f:
        pickle.dump(experts, f)
    with open(os.path.join(root, "datasets", "features", "cnn_feat_%s.p" % cfg["cnn_feat"]), "wb") as f:
        pickle.dump((feats, {"cnn_feat_dim": cnn_dim, "synthetic": True}), f)
    return train
