#!/usr/bin/env python3
"""Round-4 fixtures: the two ego_forecast env branches round 3 still refused, and a device-sized state-regression head.

Imports the reference from /root/reference exactly as tools/gen_golden.py does (same stubs for the absent third-party
modules, nothing copied) and records, in float64:

    tests/golden/obs_phase.npz       HumanoidEnv.get_full_obs (ego_pose/envs/humanoid_v1.py:73-96) with cfg.obs_phase: the extra
                                     last column min(cur_t / env_episode_len, 1) (:92-94), for cur_t below, at and beyond the
                                     episode length, under the default options and under one non-default combination.
    tests/golden/random_cur_t.npz    MujocoEnv.reset -> HumanoidEnv.reset_model (envs/common/mujoco_env.py:84-93,
                                     humanoid_v1.py:206-233) with cfg.random_cur_t (:218-220) on a duck-typed env, then
                                     HumanoidEnv.step (:179-199) with the simulator stubbed out, until `end`: per episode the
                                     take, start_ind, the random cur_t, the state that was set (expert frame start_ind + cur_t),
                                     the number of steps to `end` (env_episode_len - cur_t), the expert index of every step
                                     (get_expert_index(cur_t) = start_ind + cur_t) and the phase column of every observation.
    tests/golden/videoreg_head.npz   VideoRegNet(no_cnn=True) (models/video_reg_net.py:10-59) at the shipped widths (cnn_fdim 128,
                                     v_hdim 128 = 2 x 64 LSTM units, MLP [300, 200], out 115): weights, a (T = 40, B = 3) feature
                                     clip and the outputs -- hidden size 64 is what the HIP LSTM kernels run, so the GPU module
                                     (HIP LSTM + HIP GEMM head) can be held against a reference-derived number.

Runs ONLY in the build container (the reference never travels to the GPU box). Own seeds.
"""
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import gen_golden as G          # noqa: E402  (stubs + workdir helpers)


def _setup():
    G.install_stubs()
    if G.REF not in sys.path:
        sys.path.insert(0, G.REF)
    G.enter_workdir()
    import torch
    torch.set_default_dtype(torch.float64)
    import utils  # noqa: F401  (reference utils)
    from ego_pose.envs import humanoid_v1 as hv1
    from egopose_amd.skeleton import load_skeleton
    sk = load_skeleton(os.path.join(G.REF, "assets/mujoco_models/humanoid_1205_v1.xml"))
    return sk, hv1


def obs_phase(sk, hv1):
    rng = np.random.RandomState(4401)
    n = 20
    qpos = G.synth_qpos(rng, sk, n)
    qvel = rng.normal(size=(n, sk.nv))
    ep_len = 90                                      # config/egoforecast/subject_03.yml: env_episode_len
    cur_t = np.array([0, 1, 2, 7, 30, 44, 45, 60, 88, 89, 90, 91, 120, 200, 3, 17, 50, 75, 89, 90])
    out = dict(qpos=qpos, qvel=qvel, cur_t=cur_t, episode_len=ep_len)
    for tag, (oh, rd, oc, ov) in (("default", (False, True, "heading", "full")), ("variant", (True, False, "root", "root"))):
        cfg = types.SimpleNamespace(obs_coord=oc, obs_heading=oh, root_deheading=rd, obs_vel=ov, obs_phase=True, env_episode_len=ep_len)
        rows = []
        for i in range(n):
            env = types.SimpleNamespace(cfg=cfg, cur_t=int(cur_t[i]), data=types.SimpleNamespace(qpos=qpos[i].copy(), qvel=qvel[i].copy()))
            rows.append(hv1.HumanoidEnv.get_full_obs(env))
        out["obs_" + tag] = np.stack(rows)
    out["variant_opts"] = np.array([1, 1, 1, 1])      # obs_heading, keep root heading, obs_coord root, obs_vel 'root'
    path = os.path.join(G.OUT, "obs_phase.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.0f kB" % (os.path.getsize(path) / 1e3), out["obs_default"].shape, out["obs_variant"].shape)


def random_cur_t(sk, hv1):
    from envs.common import mujoco_env as menv
    HumanoidEnv = hv1.HumanoidEnv
    rng = np.random.RandomState(4402)
    n_takes, L, ep_len, margin = 3, 160, 24, 10
    takes = []
    for k in range(n_takes):
        q = G.synth_qpos(rng, sk, L)
        takes.append({"qpos": q, "qvel": rng.normal(size=(L, sk.nv)), "len": L, "head_height_lb": 0.0})
    cfg = types.SimpleNamespace(fr_margin=margin, env_episode_len=ep_len, env_start_first=False, random_cur_t=True, env_init_noise=0.0,
                                obs_coord="heading", obs_heading=False, root_deheading=True, obs_vel="full", obs_phase=True,
                                sync_exp_interval=100)
    n_ep = 12
    rec = dict(expert_ind=[], start_ind=[], cur_t0=[], set_qpos=[], set_qvel=[], n_steps=[], step_index=[], step_phase=[], first_phase=[])
    for ep in range(n_ep):
        env = types.SimpleNamespace()
        env.cfg = cfg
        env.fix_start_state = env.fix_expert_ind = env.fix_start_ind = env.fix_len = env.fix_head_lb = None
        env.expert_list, env.expert_arr = ["take_%d" % k for k in range(n_takes)], takes
        env.np_random = np.random.RandomState(100 + ep)
        np.random.seed(7000 + ep)                    # humanoid_v1.py:219 draws cur_t from the GLOBAL numpy generator
        env.model = types.SimpleNamespace(nq=sk.nq, nv=sk.nv)
        env.data = types.SimpleNamespace(qpos=np.zeros(sk.nq), qvel=np.zeros(sk.nv))
        env.sim = types.SimpleNamespace(reset=lambda: None)
        env.viewer, env._viewers, env.frame_skip = None, {}, 15
        state = {}

        def set_state(qp, qv, env=env, state=state):
            env.data.qpos, env.data.qvel = qp.copy(), qv.copy()
            state["q"], state["v"] = qp.copy(), qv.copy()
        env.set_state = set_state
        env.set_expert = lambda i, env=env: HumanoidEnv.set_expert(env, i)
        env.get_body_quat = lambda: np.zeros(84)
        env.sync_expert = lambda: None
        env.get_obs = lambda env=env: HumanoidEnv.get_full_obs(env)
        env.get_expert_index = lambda t, env=env: HumanoidEnv.get_expert_index(env, t)
        env.reset_model = lambda env=env: HumanoidEnv.reset_model(env)
        env.viewer_setup = lambda mode: None
        env.do_simulation = lambda a, n: None
        env.get_body_com = lambda name: np.array([0.0, 0.0, 10.0])      # the head never drops: no `fail`
        env.bquat = np.zeros(84)
        ob0 = menv.MujocoEnv.reset(env)               # cur_t = 0; reset_model then overwrites it (random_cur_t)
        rec["expert_ind"].append(env.expert_ind); rec["start_ind"].append(env.start_ind); rec["cur_t0"].append(env.cur_t)
        rec["set_qpos"].append(state["q"]); rec["set_qvel"].append(state["v"]); rec["first_phase"].append(ob0[-1])
        idx, ph = [], []
        for s in range(2 * ep_len):
            ob, r, done, info = HumanoidEnv.step(env, np.zeros(52))
            idx.append(env.get_expert_index(env.cur_t)); ph.append(ob[-1])
            assert not info["fail"]
            if done:
                assert info["end"]
                break
        rec["n_steps"].append(len(idx))
        rec["step_index"].append(np.pad(np.array(idx), (0, ep_len - len(idx)), constant_values=-1))
        rec["step_phase"].append(np.pad(np.array(ph), (0, ep_len - len(ph)), constant_values=-1.0))
    out = {k: np.asarray(v) for k, v in rec.items()}
    out.update(episode_len=ep_len, fr_margin=margin, take_len=L,
               takes_qpos=np.stack([t["qpos"] for t in takes]), takes_qvel=np.stack([t["qvel"] for t in takes]))
    assert (out["n_steps"] == ep_len - out["cur_t0"]).all() and len(set(out["cur_t0"].tolist())) > 4
    path = os.path.join(G.OUT, "random_cur_t.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.0f kB" % (os.path.getsize(path) / 1e3), "cur_t0", out["cur_t0"].tolist())


def videoreg_head():
    import torch
    from models.video_reg_net import VideoRegNet
    torch.manual_seed(4403)
    net = VideoRegNet(115, 128, 128, no_cnn=True, mlp_dim=(300, 200), v_net_type="lstm")
    net.eval()
    rng = np.random.RandomState(4403)
    x = torch.tensor(rng.normal(size=(40, 3, 128)))
    with torch.no_grad():
        y = net(x).numpy()                            # (T * B, 115), row t * B + b
    path = os.path.join(G.OUT, "videoreg_head.npz")
    np.savez_compressed(path, x=x.numpy().astype(np.float32), y=y, **{"sd_" + k: v.numpy().astype(np.float32) for k, v in net.state_dict().items()})
    # (weights and input stored in float32 -- what the GPU module computes in; y is the reference's float64 forward of the float64 module:
    #  the rounding of the stored copies is part of the test's tolerance, stated there)
    print("wrote", path, "%.0f kB" % (os.path.getsize(path) / 1e3), y.shape)


if __name__ == "__main__":
    sk, hv1 = _setup()
    obs_phase(sk, hv1)
    random_cur_t(sk, hv1)
    videoreg_head()
