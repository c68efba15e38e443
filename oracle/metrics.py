"""ORACLE (test infrastructure): eval pose metrics (north-star "joint-angle error").

Restates /root/reference/ego_pose/utils/metrics.py:5-36. Pinned against tests/golden/metrics.npz.
"""
import numpy as np

from . import quat as Q
from . import humanoid as H


def joint_angles(poses):
    poses = np.asarray(poses, float)
    root = Q.euler_from_quat_sxyz(poses[:, 3:7])
    root[:, 2] = 0.0
    return np.hstack([root, poses[:, 7:]])


def joint_vels(poses, dt):
    poses = np.asarray(poses, float)
    return H.qvel_fd(poses[:-1], poses[1:], dt, "heading")


def joint_accels(vels, dt):
    return np.diff(np.asarray(vels, float), axis=0) / dt


def mean_dist(x, y):
    return np.linalg.norm(np.asarray(x) - np.asarray(y), axis=1).mean()


def mean_abs(x):
    return np.abs(x).mean()
