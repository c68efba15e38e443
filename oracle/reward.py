"""ORACLE (test infrastructure, not product code): the quat_v3 imitation reward.

Batch-first numpy restatement of ``quat_space_reward_v3``
(/root/reference/ego_pose/core/reward_function.py:4-60) over the drained per-env state,
pinned against tests/golden/reward.npz.
"""
import numpy as np

from . import quat as Q
from . import humanoid as H

# defaults of reward_function.py:8-12
DEFAULT_WEIGHTS = dict(w_p=0.5, w_v=0.1, w_e=0.2, w_rp=0.1, w_rv=0.1, k_p=2.0, k_v=0.005, k_e=20.0,
                       k_rh=300.0, k_rq=300.0, k_rl=5.0, k_ra=0.5, v_ord=2, decay=False)


def resolve_weights(ws):
    out = dict(DEFAULT_WEIGHTS)
    if ws:
        out.update(ws)
    return out


def quat_v3(cur_qpos, prev_qpos, prev_bquat, ee_wpos, t, expert_row, weights, b_diffw, dt,
            episode_len, end, end_reward, skel_start, skel_ndof, obs_coord="heading"):
    """Returns (reward (B,), c_info (B,5)).

    ``obs_coord`` = cfg.obs_coord: the frame of the learner's root linear velocity and end-effector offsets
    (reward_function.py:19,23: get_qvel_fd(..., cfg.obs_coord), env.get_ee_pos(cfg.obs_coord)); the expert rows were
    written by gen_expert.py, which always uses 'heading' (gen_expert.py:18-22).

    ``expert_row`` is a dict of the expert table rows at index start_ind + t:
    qpos(…,59), rlinv_local(…,3), rangv(…,3), rq_rmh(…,4), ee_pos(…,15), bquat(…,84), bangvel(…,63).
    ``t`` is env.cur_t AFTER the step (reward_function.py:15-16).
    """
    ws = resolve_weights(weights)
    cur_qpos = np.atleast_2d(np.asarray(cur_qpos, float))
    prev_qpos = np.atleast_2d(np.asarray(prev_qpos, float))
    B = cur_qpos.shape[0]
    t = np.broadcast_to(np.asarray(t, float), (B,))
    end = np.broadcast_to(np.asarray(end, bool), (B,))
    # learner features (reward_function.py:18-26)
    cur_qvel = H.qvel_fd(prev_qpos, cur_qpos, dt, obs_coord)
    cur_rq_rmh = Q.de_heading(cur_qpos[:, 3:7])
    cur_ee = H.ee_pos(cur_qpos, ee_wpos, obs_coord)
    cur_bquat = H.body_quat(cur_qpos, skel_start, skel_ndof)
    cur_bangvel = H.angvel_fd(prev_bquat, cur_bquat, dt)
    e = {k: np.atleast_2d(np.asarray(v, float)) for k, v in expert_row.items()}
    # pose term (:35-38)
    pose_diff = Q.multi_quat_norm(Q.multi_quat_diff(cur_bquat[:, 4:], e["bquat"][:, 4:])) * b_diffw
    pose_r = np.exp(-ws["k_p"] * np.sum(pose_diff ** 2, axis=1))
    # body angular velocity term (:40-41)
    dv = cur_bangvel[:, 3:] - e["bangvel"][:, 3:]
    vel_dist = np.linalg.norm(dv, ord=ws["v_ord"], axis=1)
    vel_r = np.exp(-ws["k_v"] * vel_dist ** 2)
    # end-effector term (:43-44)
    ee_r = np.exp(-ws["k_e"] * np.sum((cur_ee - e["ee_pos"]) ** 2, axis=1))
    # root pose term (:46-48)
    dh = cur_qpos[:, 2] - e["qpos"][:, 2]
    dq = Q.multi_quat_norm(Q.multi_quat_diff(cur_rq_rmh, e["rq_rmh"]))[:, 0]
    rp_r = np.exp(-ws["k_rh"] * dh ** 2 - ws["k_rq"] * dq ** 2)
    # root velocity term (:50-52)
    dl = np.sum((cur_qvel[:, :3] - e["rlinv_local"]) ** 2, axis=1)
    da = np.sum((cur_qvel[:, 3:6] - e["rangv"]) ** 2, axis=1)
    rv_r = np.exp(-ws["k_rl"] * dl - ws["k_ra"] * da)
    # blend (:54-59)
    wsum = ws["w_p"] + ws["w_v"] + ws["w_e"] + ws["w_rp"] + ws["w_rv"]
    r = (ws["w_p"] * pose_r + ws["w_v"] * vel_r + ws["w_e"] * ee_r + ws["w_rp"] * rp_r + ws["w_rv"] * rv_r) / wsum
    if ws.get("decay", False):
        r = r * (1.0 - t / episode_len)
    r = r + np.where(end, end_reward, 0.0)
    return r, np.stack([pose_r, vel_r, ee_r, rp_r, rv_r], axis=1)


def constant(end, end_reward):
    """constant_reward (ego_pose/core/reward_function.py:63-67): the function builds 1 + end_reward at an episode's end
    and then RETURNS 1.0 with a one-element zero c_info."""
    return 1.0, np.zeros(1)


def pose_dist(qpos, expert_qpos, end, end_reward):
    """pose_dist_reward (reward_function.py:70-75) with HumanoidEnv.get_pose_dist (humanoid_v1.py:275-280)."""
    d = float(np.linalg.norm((np.asarray(expert_qpos, float) - np.asarray(qpos, float))[2:]))
    r = 5.0 - 3.0 * d
    if end:
        r += end_reward
    return r, np.array([d])
