#!/usr/bin/env python3
"""Round-3 fixtures: the env options the round-2 review found accepted but not honoured.

Imports the reference from /root/reference exactly as tools/gen_golden.py does (same stubs for the absent third-party
modules, nothing copied) and records, in float64:

    tests/golden/reward_root.npz     quat_space_reward_v3 (ego_pose/core/reward_function.py:4-60) with cfg.obs_coord = 'root':
                                     the learner's root linear velocity and end-effector offsets are then expressed in the
                                     ROOT frame (reward_function.py:19,23 -> utils/math.py:20-35, humanoid_v1.py:98-111) while
                                     the expert rows keep gen_expert.py's 'heading' frame (gen_expert.py:18-22 hard-codes it).
                                     Also the learner features themselves (get_qvel_fd(..., 'root'), get_ee_pos('root')) for
                                     the pose-feature kernel K7.
    tests/golden/do_simulation.npz   HumanoidEnv.do_simulation (humanoid_v1.py:158-177) under action_type 'position' and
                                     'torque': the clipped controls the reference writes into data.ctrl on each substep
                                     (sim.step replaced by a recorder that moves the state a little so substeps differ).

Runs ONLY in the build container (the reference never travels to the GPU box). Own seeds.
"""
import os
import sys
import types

import numpy as np
import yaml

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
    import utils.transformation, utils.math  # noqa: F401,E401
    from ego_pose.envs import humanoid_v1 as hv1
    from ego_pose.utils.egomimic_config import Config
    from egopose_amd.skeleton import load_skeleton
    sk = load_skeleton(os.path.join(G.REF, "assets/mujoco_models/humanoid_1205_v1.xml"))
    cfg = Config("subject_03", create_dirs=False)
    return sk, cfg, hv1, sys.modules["utils.transformation"], sys.modules["utils.math"]


def reward_root(sk, cfg, hv1, T, rmath):
    from ego_pose.core.reward_function import quat_space_reward_v3
    HumanoidEnv = hv1.HumanoidEnv
    rng = np.random.RandomState(31337)
    env = G.make_fake_env(sk, cfg, HumanoidEnv)
    L = 64
    base = G.synth_qpos(rng, sk, 1)[0]
    e_qpos = np.zeros((L, sk.nq))
    ph = rng.uniform(0, 2 * np.pi, size=sk.nq)
    fr = rng.uniform(0.5, 2.0, size=sk.nq)
    yaw0 = rng.uniform(-np.pi, np.pi)
    for f in range(L):
        tt = f / 30.0
        e_qpos[f, :2] = base[:2] + np.array([0.6 * tt, 0.2 * np.sin(tt)])
        e_qpos[f, 2] = 0.9 + 0.02 * np.sin(2 * tt)
        qy = T.quaternion_about_axis(yaw0 + 0.4 * tt, [0, 0, 1])
        qt = T.quaternion_about_axis(0.25 * np.sin(3 * tt) + 0.2, [1, 0.3, 0])      # a real tilt: root != heading frame
        e_qpos[f, 3:7] = T.quaternion_multiply(qy, qt)
        e_qpos[f, 7:] = np.clip(base[7:] + 0.2 * np.sin(fr[7:] * tt + ph[7:]), sk.joint_range[:, 0], sk.joint_range[:, 1])
    e_qpos[:, 32:35] = 0.0
    e_qpos[:, 42:45] = 0.0
    dt = env.dt
    # expert rows as gen_expert.py writes them: its own config says obs_coord 'heading' whatever the training config says
    expert = {k: [] for k in ['rlinv_local', 'rangv', 'rq_rmh', 'ee_pos', 'bquat', 'bangvel', 'qvel']}
    for f in range(L):
        env.data.qpos = e_qpos[f].copy()
        env.data.body_xpos = np.vstack([np.zeros(3), sk.body_xpos(e_qpos[f])])
        expert['rq_rmh'].append(rmath.de_heading(e_qpos[f, 3:7]))
        expert['ee_pos'].append(HumanoidEnv.get_ee_pos(env, 'heading'))
        expert['bquat'].append(HumanoidEnv.get_body_quat(env))
        if f > 0:
            qv = rmath.get_qvel_fd(e_qpos[f - 1], e_qpos[f], dt)
            expert['qvel'].append(qv)
            expert['rlinv_local'].append(rmath.transform_vec(qv[:3].copy(), e_qpos[f, 3:7], 'heading'))
            expert['rangv'].append(qv[3:6].copy())
            expert['bangvel'].append(rmath.get_angvel_fd(expert['bquat'][f - 1], expert['bquat'][f], dt))
    for k in ['qvel', 'rlinv_local', 'rangv', 'bangvel']:
        expert[k].insert(0, expert[k][0].copy())
    expert = {k: np.vstack(v) for k, v in expert.items()}
    expert['qpos'] = e_qpos
    env.expert = expert

    n = 128
    cases = dict(cur_qpos=[], prev_qpos=[], prev_bquat=[], ee_wpos=[], t=[], start_ind=[], end=[], wset=[], end_reward=[],
                 reward=[], c_info=[], reward_heading=[], c_info_heading=[], learner_qvel_root=[], learner_ee_root=[])
    wsets = [dict(cfg.reward_weights), {}, dict(cfg.reward_weights, decay=True, w_v=0.1, v_ord=2)]
    for i in range(n):
        start = int(rng.randint(0, L - 40))
        t = int(rng.randint(1, 30))
        ind = start + t
        noise = 0.0 if i % 8 == 0 else (0.02 if i % 2 else 0.15)
        prev, cur = e_qpos[ind - 1].copy(), e_qpos[ind].copy()
        for q in (prev, cur):
            q[:3] += rng.normal(size=3) * noise * 0.3
            q[3:7] = T.quaternion_multiply(q[3:7], T.quaternion_about_axis(rng.normal() * noise, rng.normal(size=3)))
            q[7:] += rng.normal(size=sk.nq - 7) * noise
        if i % 16 == 4:
            cur = prev.copy()
        if i % 16 == 5:
            cur[3:7] *= -1.0
        env.data.qpos = prev.copy()
        prev_bquat = HumanoidEnv.get_body_quat(env)
        env.data.qpos = cur.copy()
        xpos = sk.body_xpos(cur) + rng.normal(size=(21, 3)) * noise * 0.05
        env.data.body_xpos = np.vstack([np.zeros(3), xpos])
        env.prev_qpos, env.prev_bquat = prev, prev_bquat
        env.cur_t, env.start_ind = t, start
        env.end_reward = float(rng.uniform(0, 5))
        wi = i % 3
        cfg.reward_weights = wsets[wi]
        end = bool(i % 5 == 0)
        cfg.obs_coord = 'root'
        r, ci = quat_space_reward_v3(env, None, None, {'end': end})
        cases['learner_qvel_root'].append(rmath.get_qvel_fd(prev, cur, dt, 'root'))
        cases['learner_ee_root'].append(HumanoidEnv.get_ee_pos(env, 'root'))
        cfg.obs_coord = 'heading'
        rh, cih = quat_space_reward_v3(env, None, None, {'end': end})
        cases['cur_qpos'].append(cur); cases['prev_qpos'].append(prev); cases['prev_bquat'].append(prev_bquat)
        cases['ee_wpos'].append(xpos[sk.ee_body].ravel()); cases['t'].append(t); cases['start_ind'].append(start)
        cases['end'].append(end); cases['wset'].append(wi); cases['end_reward'].append(env.end_reward)
        cases['reward'].append(r); cases['c_info'].append(ci)
        cases['reward_heading'].append(rh); cases['c_info_heading'].append(cih)
    cfg.reward_weights = wsets[0]
    out = {k: np.array(v) for k, v in cases.items()}
    # the two frames must actually differ on these inputs (tilted roots), otherwise the fixture pins nothing
    assert np.abs(out['c_info'][:, 2] - out['c_info_heading'][:, 2]).max() > 1e-3
    assert np.abs(out['c_info'][:, 4] - out['c_info_heading'][:, 4]).max() > 1e-3
    path = os.path.join(G.OUT, "reward_root.npz")
    np.savez_compressed(path, **out, **{"expert_" + k: v for k, v in expert.items()}, episode_len=cfg.env_episode_len, dt=dt,
                        wset_json=np.array([yaml.safe_dump(w) for w in wsets]))
    print("wrote", path, "%.0f kB" % (os.path.getsize(path) / 1e3))


def do_simulation(sk, cfg, hv1, T, rmath):
    HumanoidEnv = hv1.HumanoidEnv
    rng = np.random.RandomState(4242)
    env = G.make_fake_env(sk, cfg, HumanoidEnv)
    env.save_video, env.viewer, env.cur_t = False, None, 0
    env.compute_torque = lambda ctrl: HumanoidEnv.compute_torque(env, ctrl)
    M0 = sk.zero_pose_inertia()
    n, n_sub = 24, 3
    qpos = G.synth_qpos(rng, sk, n)
    qvel = rng.normal(size=(n, sk.nv)) * 2.0
    action = rng.normal(size=(n, sk.nu)) * 0.6
    action[:, ::7] *= 400.0                      # far beyond the torque limits in 'torque' mode: the clip is exercised
    qMs, Cs = [], []
    out = {}
    for mode in ("position", "torque"):
        cfg.action_type = mode
        ctrls = np.zeros((n, n_sub, sk.nu))
        for i in range(n):
            d = 1.0 + 0.2 * np.random.RandomState(100 + i).uniform(-1, 1, size=sk.nv)
            M = M0 * d[:, None] * d[None, :]
            qM = sk.sparse_from_full(M)
            M = sk.full_from_sparse(qM)
            C = np.random.RandomState(200 + i).normal(size=sk.nv) * 20.0
            if mode == "position":
                qMs.append(qM); Cs.append(C)
            env.data = G.FakeData()
            env.data.qpos, env.data.qvel = qpos[i].copy(), qvel[i].copy()
            env.data.qM, env.data.qfrc_bias = qM, C
            env.data.ctrl = np.zeros(sk.nu)

            def fake_fullM(model, dst, qM_, _M=M):
                dst[:] = _M.ravel()
            hv1.mjf.mj_fullM = fake_fullM
            rec = []

            def step(_rec=rec, _d=env.data):
                _rec.append(_d.ctrl.copy())
                # a deterministic nudge of the state so that consecutive substeps see different inputs (recorded below)
                _d.qvel = _d.qvel + 0.01 * np.cos(np.arange(_d.qvel.size) + len(_rec))
                _d.qpos = _d.qpos.copy()
                _d.qpos[7:] += 0.002 * np.sin(np.arange(_d.qpos.size - 7) + len(_rec))
            env.sim = types.SimpleNamespace(step=step)
            HumanoidEnv.do_simulation(env, action[i], n_sub)
            ctrls[i] = np.stack(rec)
        out["ctrl_" + mode] = ctrls
    cfg.action_type = "position"
    # the state each substep saw (same nudges, replayed)
    states_q = np.zeros((n, n_sub, sk.nq)); states_v = np.zeros((n, n_sub, sk.nv))
    for i in range(n):
        q, v = qpos[i].copy(), qvel[i].copy()
        for s in range(n_sub):
            states_q[i, s], states_v[i, s] = q, v
            v = v + 0.01 * np.cos(np.arange(v.size) + s + 1)
            q = q.copy(); q[7:] += 0.002 * np.sin(np.arange(q.size - 7) + s + 1)
    path = os.path.join(G.OUT, "do_simulation.npz")
    np.savez_compressed(path, qpos=states_q, qvel=states_v, action=action, qM=np.stack(qMs), C=np.stack(Cs), dt=sk.timestep, **out)
    print("wrote", path, "%.0f kB" % (os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    args = _setup()
    reward_root(*args)
    do_simulation(*args)
