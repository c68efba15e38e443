"""Expert table handling: building the per-frame features the reward reads and keeping them in HBM.

The reference precomputes ``datasets/features/expert_<id>.p`` offline with
/root/reference/ego_pose/data_process/gen_expert.py:28-83 (MuJoCo FK + utils/math.py helpers) and
HumanoidEnv.load_experts unpickles it (ego_pose/envs/humanoid_v1.py:47-54). Here the same per-take dict is
either unpickled unchanged, or derived from a qpos sequence with the K7 feature kernel (GPU) + the physics
backend's forward kinematics; ``ExpertSet`` concatenates the takes, uploads the packed reward rows
(egp_upload_experts) and keeps the CNN features as one device table for gather-built LSTM windows.
"""
from __future__ import annotations

import numpy as np
import torch

HOT_KEYS = ("qpos", "qvel", "rlinv_local", "rangv", "rq_rmh", "ee_pos", "bquat", "bangvel")


def body_positions(physics, qpos_seq):
    """World body positions (L, nbody, 3) of a qpos sequence via the physics backend's FK (env 0)."""
    sk = physics.skel
    out = np.empty((qpos_seq.shape[0], len(sk.body_names), 3))
    zero_v = np.zeros(sk.nv)
    for i, q in enumerate(qpos_seq):
        physics.reset(0, q, zero_v)
        out[i] = physics.drain(0)[4]
    return out


def build_expert_take(ctx, physics, qpos_seq, lb=0, ub=None):
    """gen_expert.get_expert: features of one take from its qpos sequence (float64, reference formats)."""
    sk = ctx.skel
    qpos = np.array(qpos_seq, dtype=np.float64)
    addr = sk.body_qposaddr()
    for hand in ("LeftHand", "RightHand"):                     # "remove noisy hand data" (gen_expert.py:37-39)
        qpos[:, slice(*addr[hand])] = 0.0
    L = qpos.shape[0]
    xpos = body_positions(physics, qpos)
    ee_w = xpos[:, sk.ee_body, :].reshape(L, 15)
    dev = torch.device("cuda", ctx.device)
    cur = torch.as_tensor(qpos, device=dev)
    prev = torch.cat([cur[:1], cur[:-1]], 0).contiguous()
    f = ctx.pose_features(cur, prev, torch.as_tensor(ee_w, device=dev), expert_convention=True)
    f = {k: v.cpu().numpy() for k, v in f.items()}
    for k in ("qvel", "rlinv_local", "rangv", "bangvel"):      # frame 0 repeats frame 1 (gen_expert.py:66-69,75)
        f[k][0] = f[k][1]
    take = dict(f)
    take["rlinv"] = f["qvel"][:, :3].copy()
    take["qpos"] = qpos
    take["ee_wpos"] = ee_w
    take["head_pos"] = xpos[:, sk.body_names.index("Head"), :].copy()
    ub = L if ub is None else ub
    for k in list(take):
        take[k] = np.ascontiguousarray(take[k][lb:ub])
    take["len"] = take["qpos"].shape[0]
    take["height_lb"] = float(take["qpos"][:, 2].min())
    take["head_height_lb"] = float(take["head_pos"][:, 2].min())
    return take


class ExpertSet:
    """All training takes, concatenated; host copies feed resets, device copies feed K2 and the LSTM."""

    def __init__(self, expert_arr, cnn_feat):
        if len(expert_arr) != len(cnn_feat) or not expert_arr:
            raise ValueError("need one cnn feature array per expert take")
        self.expert_arr = expert_arr
        self.cnn_feat = cnn_feat
        self.lens = np.array([int(e["len"]) if "len" in e else int(e["qpos"].shape[0]) for e in expert_arr], np.int64)
        self.take_offset = np.concatenate([[0], np.cumsum(self.lens)]).astype(np.int64)
        self.qpos = np.ascontiguousarray(np.concatenate([e["qpos"] for e in expert_arr], 0), dtype=np.float64)
        self.qvel = np.ascontiguousarray(np.concatenate([e["qvel"] for e in expert_arr], 0), dtype=np.float64)
        self.head_height_lb = np.array([float(e["head_height_lb"]) for e in expert_arr])
        self.cnn_lens = np.array([c.shape[0] for c in cnn_feat], np.int64)
        self.cnn_offset = np.concatenate([[0], np.cumsum(self.cnn_lens)]).astype(np.int64)
        self._cnn_table = {}

    def upload(self, ctx):
        ctx.upload_experts(self.expert_arr)

    def cnn_table(self, device, dtype):
        key = (str(device), dtype)
        if key not in self._cnn_table:
            tab = np.ascontiguousarray(np.concatenate(self.cnn_feat, 0))
            self._cnn_table[key] = torch.as_tensor(tab, dtype=dtype, device=device)
        return self._cnn_table[key]
