"""Pose metrics of the evaluation path (the north star's "joint-angle error"), host side, vectorised NumPy.

Mirrors /root/reference/ego_pose/utils/metrics.py:5-36 (get_joint_angles / get_joint_vels / get_joint_accels /
get_mean_dist / get_mean_abs), the aggregation loop of ego_pose/eval_pose.py:31-69 (compute_metrics) and the two
trajectory helpers the eval scripts use (utils/tools.py:71-75 align_human_state, ego_pose/utils/tools.py:35-40
remove_noisy_hands). Quaternions are (w, x, y, z) as everywhere in the reference (utils/transformation.py).
These run once per evaluated take on a few thousand frames; they are not on the rollout hot path.
"""
from __future__ import annotations

import numpy as np


# ------------------------------------------------------------------ quaternion helpers (batched over the first axis)
def _qmul(a, b):
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1)


def _qinv(q):
    c = q * np.array([1.0, -1.0, -1.0, -1.0])
    return c / np.sum(q * q, -1, keepdims=True)              # quaternion_inverse divides by q.q (transformation.py:1421)


def _rot_matrix(q):
    """quaternion_matrix (transformation.py:1281-1291): renormalises, identity below eps."""
    q = np.asarray(q, float)
    n = np.sum(q * q, -1)
    out = np.tile(np.eye(3), q.shape[:-1] + (1, 1))
    ok = n >= np.finfo(float).eps * 4.0
    s = np.where(ok, np.sqrt(2.0 / np.where(ok, n, 1.0)), 0.0)[..., None]
    p = q * s
    w, x, y, z = p[..., 0], p[..., 1], p[..., 2], p[..., 3]
    m = np.stack([np.stack([1.0 - y * y - z * z, x * y - z * w, x * z + y * w], -1),
                  np.stack([x * y + z * w, 1.0 - x * x - z * z, y * z - x * w], -1),
                  np.stack([x * z - y * w, y * z + x * w, 1.0 - x * x - y * y], -1)], -2)
    out[ok] = m[ok]
    return out


def _heading_q(q):
    """get_heading_q (utils/math.py:62-68): keep the rotation about z only."""
    h = np.array(q, float, copy=True)
    h[..., 1] = 0.0
    h[..., 2] = 0.0
    return h / np.linalg.norm(h, axis=-1, keepdims=True)


def _euler_sxyz(q):
    """euler_from_quaternion(q, 'sxyz') = euler_from_matrix(quaternion_matrix(q)) (transformation.py:1125-1191)."""
    M = _rot_matrix(q)
    cy = np.sqrt(M[..., 0, 0] ** 2 + M[..., 1, 0] ** 2)
    big = cy > np.finfo(float).eps * 4.0
    ax = np.where(big, np.arctan2(M[..., 2, 1], M[..., 2, 2]), np.arctan2(-M[..., 1, 2], M[..., 1, 1]))
    ay = np.arctan2(-M[..., 2, 0], cy)
    az = np.where(big, np.arctan2(M[..., 1, 0], M[..., 0, 0]), 0.0)
    return np.stack([ax, ay, az], -1)


def _rotation_from_quaternion(q):
    """rotation_from_quaternion(q, separate=True) (utils/math.py): axis, angle; zero rotation -> axis x, angle 0."""
    w = q[..., 0]
    tiny = 1.0 - w * w < 1e-8
    s = np.sqrt(np.where(tiny, 1.0, 1.0 - w * w))
    axis = np.where(tiny[..., None], np.array([1.0, 0.0, 0.0]), q[..., 1:] / s[..., None])
    angle = np.where(tiny, 0.0, 2.0 * np.arccos(np.clip(w, -1.0, 1.0)))
    return axis, angle


def get_qvel_fd(cur_qpos, next_qpos, dt, transform=None):
    """utils/math.py:20-35 for whole trajectories: rows of cur_qpos / next_qpos are paired."""
    cur, nxt = np.atleast_2d(np.asarray(cur_qpos, float)), np.atleast_2d(np.asarray(next_qpos, float))
    v = (nxt[:, :3] - cur[:, :3]) / dt
    axis, angle = _rotation_from_quaternion(_qmul(nxt[:, 3:7], _qinv(cur[:, 3:7])))
    angle = np.where(angle > np.pi, angle - 2 * np.pi, np.where(angle < -np.pi, angle + 2 * np.pi, angle))
    rv = axis * (angle / dt)[:, None]
    R = _rot_matrix(cur[:, 3:7])
    rv = np.einsum("nji,nj->ni", R, rv)                       # transform_vec(.., 'root'): R^T v
    if transform is not None:
        if transform == "root":
            Rt = R
        elif transform == "heading":
            Rt = _rot_matrix(_heading_q(cur[:, 3:7]))
        else:
            raise AssertionError("unknown transform %r" % (transform,))
        v = np.einsum("nji,nj->ni", Rt, v)
    return np.hstack([v, rv, (nxt[:, 7:] - cur[:, 7:]) / dt])


# ------------------------------------------------------------------ ego_pose/utils/metrics.py
def get_joint_angles(poses):
    poses = np.asarray(poses, float)
    root = _euler_sxyz(poses[:, 3:7])
    root[:, 2] = 0.0                                         # yaw is not part of the pose error
    return np.hstack((root, poses[:, 7:]))


def get_joint_vels(poses, dt):
    poses = np.asarray(poses, float)
    return get_qvel_fd(poses[:-1], poses[1:], dt, "heading")


def get_joint_accels(vels, dt):
    return np.diff(np.asarray(vels, float), axis=0) / dt


def get_mean_dist(x, y):
    return np.linalg.norm(np.asarray(x) - np.asarray(y), axis=1).mean()


def get_mean_abs(x):
    return np.abs(x).mean()


# ------------------------------------------------------------------ ego_pose/eval_pose.py:31-69
def compute_metrics(results, dt=1.0 / 30.0, algo=None, verbose=False):
    """Mean over takes of (pose distance, velocity distance, mean |acceleration|) -- the three numbers eval_pose prints.
    `results` = {'traj_pred': {take: (T,59)}, 'traj_orig': {take: (T,59)}} as written by the eval drivers."""
    if results is None:
        return None
    per_take, acc = {}, np.zeros(3)
    for take, traj in results["traj_pred"].items():
        gt = results["traj_orig"][take]
        angs_gt, vels_gt = get_joint_angles(gt), get_joint_vels(gt, dt)
        angs, vels = get_joint_angles(traj), get_joint_vels(traj, dt)
        row = np.array([get_mean_dist(angs, angs_gt), get_mean_dist(vels, vels_gt), get_mean_abs(get_joint_accels(vels, dt))])
        per_take[take] = row
        acc += row
    acc /= max(1, len(per_take))
    out = {"pose_dist": float(acc[0]), "vel_dist": float(acc[1]), "accels": float(acc[2]), "per_take": per_take}
    if verbose:
        print("=" * 10 + " %s " % (algo or "") + "=" * 10)
        print("all - pose dist: %.4f, vel dist: %.4f, accels: %.4f" % tuple(acc))
    return out


# ------------------------------------------------------------------ trajectory helpers used by the eval drivers
def align_human_state(qpos, qvel, ref_qpos):
    """utils/tools.py:71-75, in place: put a predicted state at ref's xy position and heading."""
    qpos[:2] = ref_qpos[:2]
    hq = _heading_q(np.asarray(ref_qpos[3:7], float))
    qpos[3:7] = _qmul(hq, np.asarray(qpos[3:7], float))
    qvel[:3] = _rot_matrix(hq) @ np.asarray(qvel[:3], float)


def remove_noisy_hands(results):
    """ego_pose/utils/tools.py:35-40: zero the wrist joints (noisy in some captures) in every trajectory, in place."""
    for traj in results.values():
        for take in traj.keys():
            traj[take][..., 32:35] = 0
            traj[take][..., 42:45] = 0
