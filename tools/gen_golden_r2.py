#!/usr/bin/env python3
"""Round-2 fixtures: the reference's AgentEgo.update_params at the hidden size the HIP LSTM kernels serve.

tests/golden/ppo_update.npz (tools/gen_golden.py, G9) uses a toy video net (hidden 4 per direction), which the
persistent HIP recurrences (hidden 64 / 128 per direction) never see. This script imports the reference from
/root/reference exactly as tools/gen_golden.py does (same stubs for the absent third-party modules, nothing copied)
and records one float64 run of `AgentEgo.update_params` (ego_pose/core/agent_ego.py:34-57 -> agents/agent_ppo.py:16-65)
with `VideoStateNet(v_hdim=128)` -- the shape of every shipped ego_mimic config -- on a ragged batch of 44 episodes:

    tests/golden/ppo_update_h128.npz, ppo_update_h128_s40.npz (24 / 40 state columns)
                                       inputs, initial parameters (float32-representable, stored as float32), the
                                       reference's values / advantages / returns / log-probs / train-mode policy input
                                       before the update and every parameter after 3 epochs

Runs ONLY in the build container (the reference never travels to the GPU box). Own seeds: it does not disturb the
random stream of tools/gen_golden.py's fixtures.
"""
import copy
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import gen_golden as G          # noqa: E402  (stubs + workdir helpers)


def main(out_name="ppo_update_h128.npz", sdim=24, seed_np=2024, seed_torch=17):
    """`sdim` = 24: the round's first fixture. `sdim` = 40 (ppo_update_h128_s40.npz): at least one 32-wide k-tile of state
    columns, the shape from which on the first MLP layer gathers [context | state] itself (egopose_amd/gemm.py: GatherMlpHead)."""
    G.install_stubs()
    sys.path.insert(0, G.REF)
    G.enter_workdir()
    import torch
    torch.set_default_dtype(torch.float64)
    import utils as _ru          # noqa: F401  (reference utils: star exports the driver relies on)
    from core.common import estimate_advantages
    from core.policy_gaussian import PolicyGaussian
    from core.critic import Value
    from models.mlp import MLP
    from models.video_state_net import VideoStateNet
    from ego_pose.core.agent_ego import AgentEgo

    rng = np.random.RandomState(seed_np)
    torch.manual_seed(seed_torch)
    adim, cdim, hdim, margin, T_ep = 6, 16, 128, 10, 20
    p_vs = VideoStateNet(cdim, hdim, margin, 'lstm', None, False)
    v_vs = VideoStateNet(cdim, hdim, margin, 'lstm', None, False)
    p_net = PolicyGaussian(MLP(sdim + hdim, [32, 24], 'relu'), adim, log_std=-1.2, fix_std=True)
    v_net = Value(MLP(sdim + hdim, [32, 24], 'relu'))
    mods = [("p_vs", p_vs), ("v_vs", v_vs), ("p", p_net), ("v", v_net)]
    with torch.no_grad():        # initial parameters exactly representable in float32: both precisions start equal
        for _, mod in mods:
            for p in mod.parameters():
                p.copy_(p.float().double())
    init_np = {"init_%s__%s" % (a, k): v.numpy().astype(np.float32) for a, mod in mods for k, v in mod.state_dict().items()}
    p_params = list(p_net.parameters()) + list(p_vs.parameters())
    v_params = list(v_net.parameters()) + list(v_vs.parameters())
    lr_p, lr_v, clip = 1e-3, 2e-3, 2.0
    opt_p = torch.optim.Adam(p_params, lr=lr_p)
    opt_v = torch.optim.Adam(v_params, lr=lr_v)
    cnn_feat = [rng.normal(size=(80, cdim)).astype(np.float32).astype(np.float64),
                rng.normal(size=(66, cdim)).astype(np.float32).astype(np.float64)]
    fenv = types.SimpleNamespace(cnn_feat=cnn_feat)
    agent = AgentEgo(env=fenv, dtype=torch.float64, device=torch.device('cpu'), running_state=None,
                     custom_reward=None, mean_action=False, render=False, num_threads=1,
                     policy_net=p_net, policy_vs_net=p_vs, value_net=v_net, value_vs_net=v_vs,
                     optimizer_policy=opt_p, optimizer_value=opt_v, opt_num_epochs=3,
                     gamma=0.95, tau=0.95, clip_epsilon=0.2, policy_grad_clip=[(p_params, clip)])
    ep_lens = [20, 20, 7, 20, 3, 20, 12, 20, 20, 1, 20, 16, 20, 9, 20, 20, 5, 20, 20, 14, 20, 2,
               20, 20, 11, 20, 18, 20, 6, 20, 20, 13, 20, 4, 20, 20, 8, 20, 19, 20, 10, 20, 20, 15]
    rows = dict(states=[], actions=[], masks=[], rewards=[], exps=[], v_metas=[])
    for L_ep in ep_lens:
        e_ind = int(rng.randint(2))
        s_ind = int(rng.randint(margin, cnn_feat[e_ind].shape[0] - T_ep - margin))
        for k in range(L_ep):
            rows['states'].append(rng.normal(size=sdim).astype(np.float32).astype(np.float64))
            rows['actions'].append((rng.normal(size=adim) * 0.5).astype(np.float32).astype(np.float64))
            rows['masks'].append(0 if k == L_ep - 1 else 1)
            rows['rewards'].append(float(np.float32(rng.uniform(0, 1))))
            rows['exps'].append(1 if rng.uniform() < 0.85 else 0)
            rows['v_metas'].append([e_ind, s_ind])
    batch = types.SimpleNamespace(**{k: np.array(v) for k, v in rows.items()})
    s_p_vs, s_v_vs, s_p, s_v = copy.deepcopy((p_vs, v_vs, p_net, v_net))
    st_t = torch.from_numpy(batch.states); ac_t = torch.from_numpy(batch.actions)
    mk_t = torch.from_numpy(batch.masks).to(torch.float64); rw_t = torch.from_numpy(batch.rewards)
    for m in (s_p_vs, s_v_vs):
        m.set_mode('train'); m.initialize((mk_t, cnn_feat, batch.v_metas))
    with torch.no_grad():
        policy_in0 = s_p_vs(st_t)
        values0 = s_v(s_v_vs(st_t))
        adv0, ret0 = estimate_advantages(rw_t, mk_t, values0, 0.95, 0.95)
        logp0 = s_p.get_log_prob(policy_in0, ac_t)
    agent.update_params(batch)
    final_np = {"final_%s__%s" % (a, k): v.detach().numpy().copy() for a, mod in mods for k, v in mod.state_dict().items()}
    out = os.path.join(G.OUT, out_name)
    np.savez_compressed(
        out, cnn_feat0=cnn_feat[0].astype(np.float32), cnn_feat1=cnn_feat[1].astype(np.float32),
        states=batch.states.astype(np.float32), actions=batch.actions.astype(np.float32), masks=batch.masks,
        rewards=batch.rewards.astype(np.float32), exps=batch.exps, v_metas=batch.v_metas,
        indices=np.asarray(s_p_vs.indices), ctx_shape=np.array(s_p_vs.cnn_feat_ctx.shape),
        policy_in0_rows=np.arange(0, len(batch.masks), 5), policy_in0=policy_in0.numpy()[::5], values0=values0.numpy(), adv0=adv0.numpy(), ret0=ret0.numpy(), logp0=logp0.numpy(),
        dims=np.array([sdim, adim, cdim, hdim, margin, T_ep]), hyper=np.array([lr_p, lr_v, clip, 3, 0.95, 0.95, 0.2, -1.2]),
        **init_np, **final_np)
    print("wrote", out, "%.0f kB" % (os.path.getsize(out) / 1e3), "N =", len(batch.masks), "episodes =", len(ep_lens))


def obs_variants():
    """tests/golden/obs_variants.npz: HumanoidEnv.get_full_obs (humanoid_v1.py:73-96) under every non-default combination
    of cfg.obs_heading / root_deheading / obs_coord / obs_vel on 24 random states (unbound method on a duck-typed env)."""
    import itertools
    from ego_pose.envs import humanoid_v1 as hv1
    from egopose_amd.skeleton import load_skeleton
    sk = load_skeleton()
    rng = np.random.RandomState(77)
    qpos = G.synth_qpos(rng, sk, 24)
    qpos[:6, 3:7] *= -1.0                                    # both signs of the quaternion (get_heading flips on z < 0)
    qvel = rng.normal(size=(24, sk.nv))
    combos = list(itertools.product([False, True], [True, False], ["heading", "root"], ["full", "root", "no"]))
    out = {"qpos": qpos, "qvel": qvel, "combos": np.array([[int(a), int(b), int(c == "root"), ["full", "root", "no"].index(d)]
                                                             for a, b, c, d in combos])}
    for k, (oh, rd, oc, ov) in enumerate(combos):
        cfg = types.SimpleNamespace(obs_coord=oc, obs_heading=oh, root_deheading=rd, obs_vel=ov, obs_phase=False, env_episode_len=200)
        rows = []
        for i in range(qpos.shape[0]):
            env = types.SimpleNamespace(cfg=cfg, cur_t=0, data=types.SimpleNamespace(qpos=qpos[i].copy(), qvel=qvel[i].copy()))
            rows.append(hv1.HumanoidEnv.get_full_obs(env))
        out["obs_%d" % k] = np.stack(rows)
    path = os.path.join(G.OUT, "obs_variants.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.0f kB" % (os.path.getsize(path) / 1e3), "%d combinations" % len(combos))


def small_rewards():
    """tests/golden/reward_simple.npz: constant_reward / pose_dist_reward (ego_pose/core/reward_function.py:63-75) with
    HumanoidEnv.get_pose_dist (humanoid_v1.py:275-280) bound to a duck-typed env, 32 cases with and without the end flag."""
    from ego_pose.envs import humanoid_v1 as hv1
    from ego_pose.core.reward_function import reward_func
    from egopose_amd.skeleton import load_skeleton
    sk = load_skeleton()
    rng = np.random.RandomState(91)
    expert_qpos = G.synth_qpos(rng, sk, 40)
    qpos = G.synth_qpos(rng, sk, 32)
    frame = rng.randint(0, 40, size=32)
    end = (rng.uniform(size=32) < 0.4).astype(np.int64)
    end_reward = 3.25
    out = dict(expert_qpos=expert_qpos, qpos=qpos, frame=frame, end=end, end_reward=end_reward)
    for name in ("constant", "pose_dist"):
        rs, cs = [], []
        for i in range(32):
            env = types.SimpleNamespace(end_reward=end_reward, expert={"qpos": expert_qpos}, cur_t=0, start_ind=int(frame[i]),
                                        data=types.SimpleNamespace(qpos=qpos[i].copy()))
            env.get_expert_index = lambda t, env=env: env.start_ind + t
            env.get_pose_dist = lambda env=env: hv1.HumanoidEnv.get_pose_dist(env)
            r, c = reward_func[name](env, None, None, {"end": bool(end[i])})
            rs.append(r); cs.append(np.asarray(c, float))
        out[name + "_reward"] = np.array(rs, float)
        out[name + "_cinfo"] = np.stack(cs)
    path = os.path.join(G.OUT, "reward_simple.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.0f kB" % (os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
    main("ppo_update_h128_s40.npz", sdim=40, seed_np=2025, seed_torch=18)
    obs_variants()
    small_rewards()
