"""Reward registry of the ego_mimic task (drop-in for ego_pose/core/reward_function.py:78-80).

On the hot path the ``quat_v3`` entry is only a marker: the batched rollout evaluates K2 for all env
slots at once (egp_reward_quat_v3). Called directly -- ``reward_func['quat_v3'](env, state, action, info)``
with a single-env facade, as eval-style callers do -- it runs the same kernel on a batch of one.
"""
from __future__ import annotations

import numpy as np


def quat_space_reward_v3(env, state, action, info):
    sim = env._one()
    import torch
    dev = torch.device("cuda", sim.ctx.device)
    sim.ctx.set_reward_weights(env.cfg.reward_weights)
    t = int(env.cur_t)
    frame = int(sim.experts.take_offset[env.expert_ind]) + env.get_expert_index(t)
    as_d = lambda a: torch.as_tensor(np.asarray(a, float).reshape(1, -1), device=dev)
    as_i = lambda v: torch.tensor([int(v)], dtype=torch.int32, device=dev)
    r, ci = sim.ctx.reward(as_d(env.data_qpos), as_d(env.prev_qpos), as_d(env.ee_wpos), as_i(t), as_i(frame),
                           as_i(bool(info["end"])), float(env.end_reward))
    return float(r.item()), ci[0].cpu().numpy()


quat_space_reward_v3.egp_kernel = "quat_v3"


def constant_reward(env, state, action, info):
    """reward_function.py:63-67 (the value it RETURNS is 1.0, end bonus or not)."""
    return 1.0, np.zeros(1)


constant_reward.egp_kernel = "constant"


def pose_dist_reward(env, state, action, info):
    """reward_function.py:70-75 on the single-env facade: the same kernel on a batch of one."""
    sim = env._one()
    import torch
    dev = torch.device("cuda", sim.ctx.device)
    frame = int(sim.experts.take_offset[env.expert_ind]) + env.get_expert_index(int(env.cur_t))
    as_i = lambda v: torch.tensor([int(v)], dtype=torch.int32, device=dev)
    r, ci = sim.ctx.reward_simple("pose_dist", torch.as_tensor(np.asarray(env.data_qpos, float).reshape(1, -1), device=dev), as_i(frame),
                                  as_i(bool(info["end"])), float(env.end_reward))
    return float(r.item()), ci[0].cpu().numpy()


pose_dist_reward.egp_kernel = "pose_dist"

reward_func = {"quat_v3": quat_space_reward_v3, "constant": constant_reward, "pose_dist": pose_dist_reward}
