"""HumanoidEnv: the reference's env object re-founded on a batch of lockstep envs.

Drop-in surface of /root/reference/ego_pose/envs/humanoid_v1.py (``HumanoidEnv(cfg)``, ``seed``,
``load_experts``, ``cnn_feat``, ``model.actuator_names``, ``observation_space/action_space``, ``dt``,
``end_reward``, ``np_random``, ``set_fix_sampling`` ...) as far as the training driver touches it
(ego_pose/ego_mimic.py:41-48,112). There is no per-env Python stepping on the hot path: the agent asks for
``env.batched(n_env, device)`` and gets a ``BatchedSim`` bundling

    EgpContext (skeleton + gains + reward weights + expert table in HBM)
    physics backend (host)  +  RolloutEngine (worker threads, pinned staging, K1)

MuJoCo is not loaded; the model tree comes from the MJCF named in the config when that file exists, else
from the packaged skeleton asset. A single-env ``reset()/step()`` facade (batch of one, same kernels) is
kept for callers such as eval scripts.
"""
from __future__ import annotations

import os
import pickle
import types

import numpy as np

from .skeleton import load_skeleton, DEFAULT_ASSET


def obs_options_of(cfg):
    from .hip import obs_options_of as f          # (hip.py imports torch: keep this module light to import)
    return f(cfg)


class _Space:
    def __init__(self, dim):
        self.shape = (int(dim),)
        self.low = -np.inf * np.ones(dim)
        self.high = np.inf * np.ones(dim)


class BatchedSim:
    """Everything device/host-resident that n_env lockstep envs share."""

    def __init__(self, env, n_env, device_index=0, n_threads=None, n_groups=1, physics=None):
        from .hip import EgpContext
        from .physics import make_physics, RolloutEngine
        from .expert import ExpertSet
        cfg = env.cfg
        self.env = env
        self.n_env = int(n_env)
        self.ctx = EgpContext(env.skel, cfg.jkp, cfg.jkd, cfg.a_ref, cfg.a_scale, cfg.torque_lim, cfg.b_diffw,
                              reward_weights=cfg.reward_weights, episode_len=cfg.env_episode_len,
                              frame_skip=env.frame_skip, device=device_index, obs_options=obs_options_of(cfg))
        self.physics = physics if physics is not None else make_physics(env.skel, self.n_env, cfg)
        # EGP_DEVICE_DYNAMICS=1: qM / qfrc_bias of every substep from K8 on the GPU instead of the backend's drain
        self.engine = RolloutEngine(self.ctx, self.physics, self.n_env, n_threads=n_threads, n_groups=n_groups,
                                    device_dynamics=os.environ.get("EGP_DEVICE_DYNAMICS", "0") == "1")
        self.experts = None
        if env.expert_arr is not None:
            self.experts = ExpertSet(env.expert_arr, env.cnn_feat)
            self.experts.upload(self.ctx)

    def close(self):
        self.engine.close()
        self.physics.close()
        self.ctx.close()


class SlotView:
    """ONE slot of the lockstep rollout behind HumanoidEnv's attribute surface, for a `custom_reward` callable that has no kernel
    (agents/agent.py:53-54 calls `custom_reward(self.env, state, action, info)`; the registry's three rewards have kernels and
    never come here). Host copies of the slot's drained state, what the reference's reward functions read: `cfg`, `dt`,
    `end_reward`, `cur_t`, `start_ind`, `expert_ind`, `expert`, `data.qpos / qvel`, `prev_qpos`, `bquat`, `prev_bquat`,
    `get_expert_index`, `get_expert_attr`, `get_ee_pos`, `get_body_quat`, `get_pose_dist`, `get_pose_diff`
    (ego_pose/envs/humanoid_v1.py:98-125,256-287; ego_pose/core/reward_function.py:4-75). A slow path by construction: one
    Python call per stepped slot and tick."""

    def __init__(self, env):
        self._env = env
        self.cfg, self.model, self.skel, self.frame_skip = env.cfg, env.model, env.skel, env.frame_skip
        self.body_qposaddr = env.body_qposaddr
        self.expert_arr, self.expert_list = env.expert_arr, env.expert_list
        self.cur_t = self.start_ind = self.expert_ind = 0
        self.expert = None
        self.data = types.SimpleNamespace(qpos=None, qvel=None)
        self.prev_qpos = self.bquat = self.prev_bquat = self._ee_w = None

    dt = property(lambda self: self._env.dt)
    end_reward = property(lambda self: self._env.end_reward)

    def load(self, cur_t, start_ind, expert_ind, qpos, qvel, prev_qpos, bquat, prev_bquat, ee_w):
        self.cur_t, self.start_ind, self.expert_ind = int(cur_t), int(start_ind), int(expert_ind)
        self.expert = self.expert_arr[self.expert_ind]
        self.data.qpos, self.data.qvel = qpos, qvel
        self.prev_qpos, self.bquat, self.prev_bquat, self._ee_w = prev_qpos, bquat, prev_bquat, ee_w
        return self

    def get_expert_index(self, t):
        return self.start_ind + t

    def get_expert_attr(self, attr, ind):
        return self.expert[attr][ind, :]

    def get_body_quat(self):
        return self.bquat

    def get_ee_pos(self, transform):
        from .metrics import _heading_q, _rot_matrix
        w = np.asarray(self._ee_w, float).reshape(-1, 3)
        if transform is None:
            return w.ravel()
        q = self.data.qpos
        if transform == "root":
            R = _rot_matrix(q[3:7])
        elif transform == "heading":
            R = _rot_matrix(_heading_q(q[3:7]))
        else:
            raise AssertionError("unknown transform %r" % (transform,))
        return ((w - q[:3]) @ R).ravel()

    def get_pose_diff(self):
        return np.abs((self.expert["qpos"][self.get_expert_index(self.cur_t), :] - self.data.qpos)[2:])

    def get_pose_dist(self):
        return np.linalg.norm((self.expert["qpos"][self.get_expert_index(self.cur_t), :] - self.data.qpos)[2:])


class HumanoidEnv:

    def __init__(self, cfg):
        self.cfg = cfg
        self.frame_skip = 15
        path = getattr(cfg, "mujoco_model_file", None)
        self.skel = load_skeleton(path if path and os.path.exists(path) else DEFAULT_ASSET)
        sk = self.skel
        self.model = types.SimpleNamespace(
            actuator_names=tuple(sk.actuator_names), body_names=("world",) + tuple(sk.body_names),
            nq=sk.nq, nv=sk.nv, nu=sk.nu, opt=types.SimpleNamespace(timestep=sk.timestep),
            _body_name2id={n: i + 1 for i, n in enumerate(sk.body_names)})
        self.body_qposaddr = sk.body_qposaddr()
        oo = obs_options_of(cfg)            # humanoid_v1.py:73-96: [heading]? ++ qpos[2:] ++ {qvel | qvel[:6] | -} ++ [phase]?
        self.obs_dim = (1 if oo["obs_heading"] else 0) + sk.nq - 2 + {"full": sk.nv, "root": 6}.get(oo["obs_vel"], 0) + (1 if oo["obs_phase"] else 0)
        self.observation_space = _Space(self.obs_dim)
        self.action_space = _Space(sk.nu)
        self.end_reward = 0.0
        self.cur_t = 0
        self.start_ind = 0
        self.expert_ind = None
        self.expert_id = None
        self.expert_list = None
        self.expert_arr = None
        self.expert = None
        self.cnn_feat = None
        self.fix_expert_ind = self.fix_start_ind = self.fix_len = self.fix_start_state = None
        self.fix_cnn_feat = self.fix_head_lb = None
        self.np_random = None
        self._sims = {}
        self._single = None
        self.seed()

    # ------------------------------------------------------------------ reference API
    @property
    def dt(self):
        return self.skel.timestep * self.frame_skip

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def load_experts(self, expert_list, expert_feat_file, cnn_feat_file):
        self.expert_list = list(expert_list)
        with open(expert_feat_file, "rb") as f:
            expert_dict = pickle.load(f)
        self.expert_arr = [expert_dict[name] for name in self.expert_list]
        with open(cnn_feat_file, "rb") as f:
            cnn_dict, _ = pickle.load(f)
        self.cnn_feat = [cnn_dict[name] for name in self.expert_list]
        self.set_expert(0)
        for sim in self._sims.values():
            sim.close()
        self._sims.clear()

    def set_experts(self, expert_list, expert_arr, cnn_feat):
        """Same as load_experts for tables that are already in memory."""
        self.expert_list, self.expert_arr, self.cnn_feat = list(expert_list), list(expert_arr), list(cnn_feat)
        self.set_expert(0)

    def set_expert(self, expert_ind):
        self.expert_ind = expert_ind
        self.expert_id = self.expert_list[expert_ind]
        self.expert = self.expert_arr[expert_ind]

    def set_fix_sampling(self, expert_ind=None, start_ind=None, len=None, start_state=None, cnn_feat=None):
        self.fix_expert_ind, self.fix_start_ind, self.fix_len = expert_ind, start_ind, len
        self.fix_start_state, self.fix_cnn_feat = start_state, cnn_feat

    def set_fix_head_lb(self, fix_head_lb=None):
        self.fix_head_lb = fix_head_lb

    def get_expert_index(self, t):
        return self.start_ind + t

    def get_expert_attr(self, attr, ind):
        return self.expert[attr][ind, :]

    def get_episode_cnn_feat(self):
        fm = self.cfg.fr_margin
        n = self.cfg.env_episode_len if self.fix_len is None else self.fix_len
        if self.fix_cnn_feat is not None:
            return self.fix_cnn_feat
        return self.cnn_feat[self.expert_ind][self.start_ind - fm: self.start_ind + n + fm, :]

    # ------------------------------------------------------------------ batched access (the hot path)
    def batched(self, n_env, device_index=0, n_threads=None, n_groups=1):
        key = (int(n_env), int(device_index), n_threads, int(n_groups))
        if key not in self._sims:
            self._sims[key] = BatchedSim(self, n_env, device_index, n_threads, n_groups)
        return self._sims[key]

    def close(self):
        for sim in self._sims.values():
            sim.close()
        self._sims.clear()
        if self._single is not None:
            self._single.close()
            self._single = None

    # ------------------------------------------------------------------ single-env facade (batch of one, same kernels)
    def _one(self):
        if self._single is None:
            self._single = BatchedSim(self, 1, 0, n_threads=1, n_groups=1)
        return self._single

    @property
    def data(self):
        """mjData-like view of the single env: qpos / qvel (host copies of the drained state)."""
        eng = self._one().engine
        return types.SimpleNamespace(qpos=eng.qpos_host[0].copy(), qvel=eng.qvel_host[0].copy())

    def reset(self):
        """MujocoEnv.reset + HumanoidEnv.reset_model (envs/common/mujoco_env.py:84-93, humanoid_v1.py:201-231)."""
        import torch
        sim = self._one()
        if self.fix_start_state is not None:
            qpos, qvel = self.fix_start_state[:self.skel.nq], self.fix_start_state[self.skel.nq:]
        else:
            e_ind, s_ind = self.sample_reset(1)
            self.set_expert(int(e_ind[0]))
            self.start_ind = int(s_ind[0])
            ind = self.start_ind
            self.cur_t = 0
            if getattr(self.cfg, "random_cur_t", False):       # humanoid_v1.py:218-220 (the reference draws from the global numpy generator)
                self.cur_t = int(self.np_random.randint(self.cfg.env_episode_len))
                ind += self.cur_t
            qpos = self.expert["qpos"][ind].copy()
            qvel = self.expert["qvel"][ind].copy()
            if self.cfg.env_init_noise > 0:
                qpos[7:] += self.np_random.normal(0.0, self.cfg.env_init_noise, size=self.skel.nq - 7)
        if self.fix_start_state is not None:
            self.cur_t = 0
        sim.engine.reset(np.array([0]), qpos[None], qvel[None])
        torch.cuda.synchronize()
        self.prev_qpos = None
        self.bquat = self.get_body_quat()
        return self.get_obs()

    def set_state(self, qpos, qvel):
        """MujocoEnv.set_state (envs/common/mujoco_env.py:95-101): overwrite the single env's state + forward."""
        import torch
        qpos, qvel = np.asarray(qpos, dtype=np.float64), np.asarray(qvel, dtype=np.float64)
        assert qpos.shape == (self.skel.nq,) and qvel.shape == (self.skel.nv,)
        sim = self._one()
        sim.engine.reset(np.array([0]), qpos[None], qvel[None])
        torch.cuda.synchronize()
        self.bquat = self.get_body_quat()

    def step(self, a):
        """HumanoidEnv.step (humanoid_v1.py:179-199) through the engine: 15 x {K1 <-> physics}."""
        import torch
        sim = self._one()
        eng = sim.engine
        self.prev_qpos = eng.qpos_host[0].copy()
        self.prev_qvel = eng.qvel_host[0].copy()
        self.prev_bquat = self.bquat.copy()
        act = torch.as_tensor(np.asarray(a, dtype=np.float64).reshape(1, -1), device=eng.qpos.device)
        eng.step_async(0, act)
        eng.wait(0)
        torch.cuda.synchronize()
        self.cur_t += 1
        self.bquat = self.get_body_quat()
        self.data_qpos = eng.qpos_host[0].copy()
        self.ee_wpos = eng.ee_wpos.cpu().numpy()[0]
        head_z = float(eng.head_z[0])
        if self.fix_head_lb is not None:
            fail = head_z < self.fix_head_lb
        else:
            fail = self.expert is not None and head_z < self.expert["head_height_lb"] - 0.1
        end = self.cur_t >= (self.cfg.env_episode_len if self.fix_len is None else self.fix_len)
        return self.get_obs(), 1.0, bool(fail or end), {"fail": bool(fail), "end": bool(end)}

    def get_obs(self):
        import torch
        sim = self._one()
        pt = torch.tensor([self.cur_t], dtype=torch.int32, device=sim.engine.qpos.device) if sim.ctx.obs_phase else None
        return sim.ctx.obs(sim.engine.qpos, sim.engine.qvel, phase_t=pt).cpu().numpy()[0]

    get_full_obs = get_obs

    def get_body_quat(self):
        sim = self._one()
        return sim.ctx.body_quat(sim.engine.qpos).cpu().numpy()[0]

    def get_ee_pos(self, transform):
        """World end-effector positions (transform None), or root-relative in the 'heading' / 'root' frame
        (humanoid_v1.py:98-111 with transform_vec, utils/math.py:47-59). Single-env facade: host arithmetic."""
        from .metrics import _heading_q, _rot_matrix
        eng = self._one().engine
        w = eng.ee_wpos.cpu().numpy()[0].reshape(-1, 3)
        if transform is None:
            return w.ravel()
        q = eng.qpos_host[0]
        if transform == "root":
            R = _rot_matrix(q[3:7])
        elif transform == "heading":
            R = _rot_matrix(_heading_q(q[3:7]))
        else:
            raise AssertionError("unknown transform %r" % (transform,))          # transform_vec's `assert False`
        return ((w - q[:3]) @ R).ravel()                                         # rows R^T (w - root)

    def sample_reset(self, n):
        """Reset sampling of reset_model (humanoid_v1.py:206-216) for n envs at once:
        expert take ~ randint(n_takes), start frame ~ randint(fr_margin, len - episode_len - fr_margin)."""
        cfg = self.cfg
        lens = np.array([int(e["len"]) for e in self.expert_arr])
        if self.fix_expert_ind is None:
            e_ind = self.np_random.randint(len(self.expert_arr), size=n)
        else:
            e_ind = np.full(n, self.fix_expert_ind)
        if self.fix_start_ind is not None:
            s_ind = np.full(n, self.fix_start_ind)
        elif getattr(cfg, "env_start_first", False):
            s_ind = np.zeros(n, dtype=np.int64)
        else:
            hi = lens[e_ind] - cfg.env_episode_len - cfg.fr_margin
            if (hi <= cfg.fr_margin).any():
                raise ValueError("expert take shorter than env_episode_len + 2*fr_margin")
            s_ind = cfg.fr_margin + (self.np_random.random_sample(n) * (hi - cfg.fr_margin)).astype(np.int64)
        return e_ind.astype(np.int64), s_ind.astype(np.int64)
