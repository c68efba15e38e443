"""ORACLE (test infrastructure): one-env-per-worker CPU humanoid env + the CPU sampler baseline.

Restates the control flow of ``HumanoidEnv`` (/root/reference/ego_pose/envs/humanoid_v1.py:158-231: reset_model,
step, do_simulation with 15 x {compute_torque; clip; sim.step}) and of ``quat_space_reward_v3`` on top of the
oracle's numpy arithmetic, one env at a time, float64 -- i.e. the per-step cost structure of the reference
sampler. Physics goes through the SAME host backend the GPU path uses (the product's physics boundary, called
through ctypes per env): MuJoCo is not available, so both sides step the deterministic surrogate.

``python -m oracle.cpu_env --threads 2 --steps 4000`` is what bench.py's ``cpu_baseline`` leg runs (in a
subprocess, OMP_NUM_THREADS=1 as the reference's README recommends).
"""
from __future__ import annotations

import argparse
import json
import os
import pickle
import sys
import time

import numpy as np
import torch

from . import humanoid as H
from . import nets as N
from . import reward as R
from . import sampler as S
from .zfilter import ZFilterOracle


class OracleHumanoidEnv:
    def __init__(self, skel, cfg, physics, expert_arr, cnn_feat, seed=0, env_slot=0, device_dynamics=False):
        """`device_dynamics`: qM / qfrc_bias are not taken from the backend's drain but evaluated by oracle/dynamics.py with the
        timing MuJoCo gives the reference (humanoid_v1.py:130-144 reads data.qM / data.qfrc_bias as the previous mj_step left them,
        i.e. at the state that step started from; sim.forward() makes them fresh after a reset, envs/common/mujoco_env.py:97-101):
        what the engine's device-dynamics mode (K8) must reproduce."""
        self.skel, self.cfg, self.phys = skel, cfg, physics
        self.device_dynamics = bool(device_dynamics)
        self._held = None
        self.slot = env_slot
        self.expert_arr, self.cnn_feat = expert_arr, cnn_feat
        self.np_random = np.random.RandomState(seed)
        self.dt = skel.timestep * 15
        self.end_reward = 0.0
        self.cur_t = 0
        self.expert_ind = 0
        self.start_ind = 0
        self.qpos = self.qvel = self.prev_qpos = self.prev_bquat = self.bquat = self.xpos = None

    def _drain(self, want_xpos):
        q, v, qM, bias, xpos = self.phys.drain(self.slot, want_xpos=want_xpos)
        self.qpos, self.qvel, self.qM, self.bias = q, v, qM, bias
        if want_xpos:
            self.xpos = xpos

    def forward(self):
        """sim.forward() after set_state: drained state, and (device dynamics) M, C evaluated at it."""
        self._drain(True)
        if self.device_dynamics:
            from . import dynamics as D
            M, C, _ = D.crba_rne_spatial(self.skel, self.qpos, self.qvel)
            self._held = (M, C)

    def _obs(self):
        c = self.cfg
        return H.full_obs(self.qpos, self.qvel, obs_heading=getattr(c, "obs_heading", False), root_deheading=getattr(c, "root_deheading", True),
                          obs_coord=getattr(c, "obs_coord", "heading"), obs_vel=getattr(c, "obs_vel", "full"),
                          phase=([self.cur_t], c.env_episode_len) if getattr(c, "obs_phase", False) else None)[0]

    def reset(self):
        cfg = self.cfg
        self.cur_t = 0
        self.expert_ind = self.np_random.randint(len(self.expert_arr))
        e = self.expert_arr[self.expert_ind]
        self.start_ind = self.np_random.randint(cfg.fr_margin, e["len"] - cfg.env_episode_len - cfg.fr_margin)
        ind = self.start_ind
        if getattr(cfg, "random_cur_t", False):                 # humanoid_v1.py:218-220
            self.cur_t = int(self.np_random.randint(cfg.env_episode_len))
            ind += self.cur_t
        self.phys.reset(self.slot, e["qpos"][ind], e["qvel"][ind])
        self.forward()
        self.bquat = H.body_quat(self.qpos, self.skel.body_qpos_start, self.skel.body_ndof)[0]
        return self._obs()

    def step(self, action):
        cfg, sk = self.cfg, self.skel
        self.prev_qpos, self.prev_bquat = self.qpos.copy(), self.bquat.copy()
        for i in range(15):
            if self.device_dynamics:
                from . import dynamics as D
                M, bias = self._held                     # left behind by the previous mj_step (or the reset's forward)
                Mn, Cn, _ = D.crba_rne_spatial(sk, self.qpos, self.qvel)
                self._held = (Mn, Cn)                    # ... and by this one: evaluated at the state it starts from
            else:
                M, bias = H.full_from_sparse(self.qM, sk.dof_parentid, sk.dof_Madr), self.bias
            _, tau = H.control_torque(getattr(cfg, "action_type", "position"), self.qpos, self.qvel, action, M, bias, cfg.jkp,
                                      cfg.jkd, cfg.a_ref, cfg.a_scale, cfg.torque_lim, sk.timestep)
            self.phys.step(self.slot, tau[0])
            self._drain(i == 14)
        self.cur_t += 1
        self.bquat = H.body_quat(self.qpos, sk.body_qpos_start, sk.body_ndof)[0]
        head_z = self.xpos[sk.body_names.index("Head"), 2]
        fail = head_z < self.expert_arr[self.expert_ind]["head_height_lb"] - 0.1
        end = self.cur_t >= cfg.env_episode_len
        return self._obs(), 1.0, bool(fail or end), {"fail": bool(fail), "end": bool(end)}

    def episode_cnn_feat(self):
        fm = self.cfg.fr_margin
        return self.cnn_feat[self.expert_ind][self.start_ind - fm: self.start_ind + self.cfg.env_episode_len + fm]

    def reward(self, state, action, info):
        e = self.expert_arr[self.expert_ind]
        ind = self.start_ind + self.cur_t
        kind = getattr(self.cfg, "reward_id", "quat_v3")
        if kind == "constant":
            return R.constant(info["end"], self.end_reward)
        if kind == "pose_dist":
            return R.pose_dist(self.qpos, e["qpos"][ind], info["end"], self.end_reward)
        row = {k: e[k][ind] for k in ("qpos", "rlinv_local", "rangv", "rq_rmh", "ee_pos", "bquat", "bangvel")}
        r, ci = R.quat_v3(self.qpos, self.prev_qpos, self.prev_bquat, self.xpos[self.skel.ee_body].ravel(), self.cur_t, row,
                          self.cfg.reward_weights, self.cfg.b_diffw, self.dt, self.cfg.env_episode_len, info["end"],
                          self.end_reward, self.skel.body_qpos_start, self.skel.body_ndof,
                          obs_coord=getattr(self.cfg, "obs_coord", "heading"))
        return float(r[0]), ci[0]


def run_sampler(skel, cfg, physics, expert_arr, cnn_feat, p_pol, p_pvs, min_batch_size, num_threads, seed=1, use_fork=True):
    """The reference's Agent.sample with AgentEgo hooks on the CPU: returns (batch, log, seconds)."""
    env = OracleHumanoidEnv(skel, cfg, physics, expert_arr, cnn_feat, seed=seed)
    rs = ZFilterOracle(skel.nq - 2 + skel.nv, clip=5)
    ctx = {}

    def pre_episode(e):
        with torch.no_grad():
            ctx["v_out"] = N.vsnet_test_init(p_pvs, e.episode_cnn_feat(), cfg.fr_margin)

    def select_action(state, t, use_mean):
        x = torch.cat((ctx["v_out"][[t]], N.as_t(state).unsqueeze(0)), dim=1)
        mean, std = N.policy_mean_std(p_pol, x)
        a = mean if use_mean else torch.normal(mean, std)
        return a[0].numpy()

    # every forked worker must step its own physics slot
    orig_reset = env.reset

    def reset_with_slot():
        return orig_reset()
    env.reset = reset_with_slot
    t0 = time.time()
    batch, log = S.sample(min_batch_size, num_threads, env, select_action, running_state=rs,
                          custom_reward=lambda e, s, a, info: e.reward(s, a, info), pre_episode=pre_episode,
                          v_meta_fn=lambda e: np.array([e.expert_ind, e.start_ind]), use_fork=use_fork)
    return batch, log, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", required=True, help="directory written by egopose_amd.synthetic.make_dataset")
    ap.add_argument("--cfg", default="subject_03")
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--update-steps", type=int, default=0,
                    help="also time the reference-structured CPU update (oracle.ppo.update_params: LSTMCell loops, 10 full-batch epochs, "
                         "float64) on the first episodes of the sample that cover this many steps; torch threads = --update-threads")
    ap.add_argument("--update-threads", type=int, default=0)
    args = ap.parse_args()
    torch.set_num_threads(1)
    torch.set_default_dtype(torch.float64)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    from egopose_amd.config import Config
    from egopose_amd.nets import MLP, PolicyGaussian, VideoStateNet
    from egopose_amd.physics import SurrogatePhysics
    from egopose_amd.skeleton import load_skeleton
    os.chdir(args.dataset)
    cfg = Config(args.cfg, create_dirs=False)
    skel = load_skeleton()
    with open(cfg.expert_feat_file, "rb") as f:
        ed = pickle.load(f)
    with open(cfg.cnn_feat_file, "rb") as f:
        cd, _ = pickle.load(f)
    takes = cfg.takes["train"]
    expert_arr, cnn_feat = [ed[t] for t in takes], [cd[t] for t in takes]
    torch.manual_seed(cfg.seed)
    np.random.seed(cfg.seed)
    cdim = cnn_feat[0].shape[-1]
    pvs = VideoStateNet(cdim, cfg.policy_v_hdim, cfg.fr_margin, "lstm", None, False)
    pol = PolicyGaussian(MLP(115 + cfg.policy_v_hdim, cfg.policy_hsize, cfg.policy_htype), 52, log_std=cfg.log_std, fix_std=cfg.fix_std)
    p_pol = {k: v.detach().clone() for k, v in pol.state_dict().items()}
    p_pvs = {k: v.detach().clone() for k, v in pvs.state_dict().items()}
    phys = SurrogatePhysics(skel, 1)
    batch, log, secs = run_sampler(skel, cfg, phys, expert_arr, cnn_feat, p_pol, p_pvs, args.steps, args.threads, seed=cfg.seed)
    out = {"env_steps": int(log.num_steps), "seconds": secs, "env_steps_per_s": log.num_steps / secs,
           "threads": args.threads, "episodes": int(log.num_episodes), "avg_c_reward": float(log.avg_c_reward),
           "physics": phys.name}
    if args.update_steps > 0:
        # AgentEgo.update_params as the reference runs it (ego_pose/core/agent_ego.py:34-57 on to_device(...) CPU tensors):
        # the update sees whole episodes, so cut at an episode end
        from . import ppo as P
        from egopose_amd.nets import Value
        ends = np.nonzero(np.asarray(batch["masks"]) == 0)[0]
        cut = int(ends[np.searchsorted(ends, min(args.update_steps, len(batch["masks"])) - 1)]) + 1
        sub = {k: v[:cut] for k, v in batch.items()}
        vvs = VideoStateNet(cdim, cfg.value_v_hdim, cfg.fr_margin, "lstm", None, False)
        val = Value(MLP(115 + cfg.value_v_hdim, cfg.value_hsize, cfg.value_htype))
        p_val = {k: v.detach().clone() for k, v in val.state_dict().items()}
        p_vvs = {k: v.detach().clone() for k, v in vvs.state_dict().items()}
        n_thr = args.update_threads or args.threads
        torch.set_num_threads(n_thr)
        t0 = time.time()
        P.update_params(dict(p_pol), dict(p_pvs), p_val, p_vvs, sub, cnn_feat, margin=cfg.fr_margin, gamma=cfg.gamma, tau=cfg.tau,
                        clip_eps=cfg.clip_epsilon, epochs=cfg.num_optim_epoch, lr_policy=cfg.policy_lr, lr_value=cfg.value_lr, grad_clip=40)
        out["update"] = {"samples": cut, "episodes": int((np.asarray(sub["masks"]) == 0).sum()), "seconds": time.time() - t0,
                         "epochs": int(cfg.num_optim_epoch), "torch_threads": n_thr}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
