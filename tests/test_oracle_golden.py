"""The oracle (oracle/*.py) against golden vectors produced by the reference itself
(tools/gen_golden.py). float64; tolerances are absolute+relative 1e-12 unless noted."""
import numpy as np
import pytest
import torch
import yaml

from conftest import load_golden
from oracle import quat as Q, humanoid as H, reward as R, gae as G, zfilter as Z, nets as N, ppo as P
from oracle import metrics as M, sampler as S

TOL = dict(rtol=1e-12, atol=1e-12)


def test_quat_known_answers():
    g = load_golden("quat.npz")
    # doctest values quoted in the reference's transformation.py
    np.testing.assert_allclose(g["kat_about_axis"], [0.99810947, 0.06146124, 0, 0], atol=1e-8)
    np.testing.assert_allclose(g["kat_mul"], [28, -44, -14, 48])
    np.testing.assert_allclose(Q.q_about_axis(0.123, [1.0, 0, 0]), g["kat_about_axis"], **TOL)
    np.testing.assert_allclose(Q.qmul([4.0, 1, -2, 3], [8.0, -5, 6, 7]), g["kat_mul"], **TOL)
    np.testing.assert_allclose(Q.qmat([0.0, 1, 0, 0]), g["kat_mat"], **TOL)
    np.testing.assert_allclose(Q.euler_from_quat_sxyz(g["kat_about_axis"]), [0.123, 0, 0], atol=1e-12)


def test_quat_random_cases():
    g = load_golden("quat.npz")
    np.testing.assert_allclose(Q.qmul(g["q1"], g["q0"]), g["mul"], **TOL)
    np.testing.assert_allclose(Q.qinv(g["q0"]), g["inv"], **TOL)
    np.testing.assert_allclose(Q.qmat(g["q0"]), g["mat"], **TOL)
    e = g["eul"]
    np.testing.assert_allclose(Q.q_from_euler_sxyz(e[:, 0], e[:, 1], e[:, 2]), g["from_euler"], **TOL)
    np.testing.assert_allclose(Q.q_about_axis(g["ang"], g["v3"]), g["about_axis"], **TOL)
    np.testing.assert_allclose(Q.rot_vec(g["qn"]), g["rot_vec"], **TOL)
    ax, an = Q.rot_axis_angle(g["qn"])
    np.testing.assert_allclose(ax, g["rot_axis"], **TOL)
    np.testing.assert_allclose(an, g["rot_angle"], **TOL)
    assert an[0] == 0.0 and an[1] == 0.0          # the 1-w<1e-8 branch
    qn = g["qn"][4:]
    np.testing.assert_allclose(Q.heading_q(qn), g["heading_q"], **TOL)
    np.testing.assert_allclose(Q.heading(qn), g["heading"], **TOL)
    np.testing.assert_allclose(Q.de_heading(qn), g["de_heading"], **TOL)
    np.testing.assert_allclose(Q.transform_vec(g["v3"], g["qn"], "root"), g["tv_root"], **TOL)
    np.testing.assert_allclose(Q.transform_vec(g["v3"][4:], qn, "heading"), g["tv_heading"], **TOL)
    np.testing.assert_allclose(Q.quat_mul_vec(g["qn"], g["v3"]), g["quat_mul_vec"], **TOL)
    np.testing.assert_allclose(Q.quat_from_expmap(g["emap"]), g["expmap"], **TOL)
    np.testing.assert_allclose(Q.euler_from_quat_sxyz(g["qn"]), g["euler_from_quat"], **TOL)
    d = Q.multi_quat_diff(g["q1"].ravel(), g["qn"].ravel())
    np.testing.assert_allclose(d, g["multi_diff"], **TOL)
    np.testing.assert_allclose(Q.multi_quat_norm(d), g["multi_norm"], **TOL)
    with pytest.raises(AssertionError):
        Q.transform_vec(g["v3"], g["qn"], "bogus")


def test_body_quat_and_obs(skel):
    g = load_golden("body_quat_obs.npz")
    np.testing.assert_allclose(H.body_quat(g["qpos"], skel.body_qpos_start, skel.body_ndof), g["bquat"], **TOL)
    np.testing.assert_allclose(H.full_obs(g["qpos"], g["qvel"]), g["obs"], **TOL)


def test_pd_torque(skel):
    g = load_golden("pd_torque.npz")
    c = load_golden("config_subject_03.npz")
    M_ = H.full_from_sparse(g["qM"], skel.dof_parentid, skel.dof_Madr)
    np.testing.assert_allclose(M_[:2], g["M"], rtol=0, atol=0)
    tau, tau_c = H.pd_torque(g["qpos"], g["qvel"], g["action"], M_, g["C"], c["jkp"], c["jkd"], c["a_ref"],
                             c["a_scale"], c["torque_lim"], float(g["dt"]))
    np.testing.assert_allclose(tau, g["torque"], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(tau_c, g["torque_clipped"], rtol=1e-11, atol=1e-10)


def _expert_rows(g, ind):
    return {k: g["expert_" + k][ind] for k in ["qpos", "rlinv_local", "rangv", "rq_rmh", "ee_pos", "bquat", "bangvel"]}


def test_reward_quat_v3(skel):
    g = load_golden("reward.npz")
    c = load_golden("config_subject_03.npz")
    wsets = [yaml.safe_load(str(s)) for s in g["wset_json"]]
    for wi, ws in enumerate(wsets):
        sel = np.where(g["wset"] == wi)[0]
        ind = g["start_ind"][sel] + g["t"][sel]
        r, ci = R.quat_v3(g["cur_qpos"][sel], g["prev_qpos"][sel], g["prev_bquat"][sel], g["ee_wpos"][sel],
                          g["t"][sel], _expert_rows(g, ind), ws, c["b_diffw"], float(g["dt"]),
                          int(g["episode_len"]), g["end"][sel], g["end_reward"][sel],
                          skel.body_qpos_start, skel.body_ndof)
        np.testing.assert_allclose(ci, g["c_info"][sel], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(r, g["reward"][sel], rtol=1e-11, atol=1e-12)
    # prev_bquat is always get_body_quat(prev_qpos) (humanoid_v1.py:184,188) -- what the fused kernel relies on
    np.testing.assert_allclose(H.body_quat(g["prev_qpos"], skel.body_qpos_start, skel.body_ndof), g["prev_bquat"], **TOL)


def test_reward_quat_v3_in_the_root_frame(skel):
    """cfg.obs_coord = 'root' inside the reward (reward_function.py:19,23): learner root velocity / end effectors in the root
    frame against expert rows that keep gen_expert.py's heading frame. The fixture also holds the same cases evaluated with
    'heading', and the two differ -- an oracle that ignored the option could not pass both."""
    g = load_golden("reward_root.npz")
    c = load_golden("config_subject_03.npz")
    wsets = [yaml.safe_load(str(s)) for s in g["wset_json"]]
    for coord, rk, ck in (("root", "reward", "c_info"), ("heading", "reward_heading", "c_info_heading")):
        for wi, ws in enumerate(wsets):
            sel = np.where(g["wset"] == wi)[0]
            ind = g["start_ind"][sel] + g["t"][sel]
            r, ci = R.quat_v3(g["cur_qpos"][sel], g["prev_qpos"][sel], g["prev_bquat"][sel], g["ee_wpos"][sel], g["t"][sel],
                              _expert_rows(g, ind), ws, c["b_diffw"], float(g["dt"]), int(g["episode_len"]), g["end"][sel],
                              g["end_reward"][sel], skel.body_qpos_start, skel.body_ndof, obs_coord=coord)
            np.testing.assert_allclose(ci, g[ck][sel], rtol=1e-11, atol=1e-12)
            np.testing.assert_allclose(r, g[rk][sel], rtol=1e-11, atol=1e-12)
    assert np.abs(g["c_info"][:, [2, 4]] - g["c_info_heading"][:, [2, 4]]).max() > 1e-3
    np.testing.assert_allclose(H.qvel_fd(g["prev_qpos"], g["cur_qpos"], float(g["dt"]), "root"), g["learner_qvel_root"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(H.ee_pos(g["cur_qpos"], g["ee_wpos"], "root"), g["learner_ee_root"], **TOL)


def test_do_simulation_control_law(skel):
    """do_simulation (humanoid_v1.py:158-177) under both action types: the clipped control of every substep."""
    g = load_golden("do_simulation.npz")
    c = load_golden("config_subject_03.npz")
    M_ = H.full_from_sparse(g["qM"], skel.dof_parentid, skel.dof_Madr)
    for mode in ("position", "torque"):
        for s in range(g["qpos"].shape[1]):
            _, tc = H.control_torque(mode, g["qpos"][:, s], g["qvel"][:, s], g["action"], M_, g["C"], c["jkp"], c["jkd"], c["a_ref"],
                                     c["a_scale"], c["torque_lim"], float(g["dt"]))
            np.testing.assert_allclose(tc, g["ctrl_" + mode][:, s], rtol=1e-11, atol=1e-10)
    assert (np.abs(g["ctrl_torque"]) == c["torque_lim"]).any() and (np.abs(g["ctrl_torque"]) < c["torque_lim"]).any()
    assert np.ptp(g["ctrl_torque"], axis=1).max() == 0.0 and np.ptp(g["ctrl_position"], axis=1).max() > 0.0
    with pytest.raises(UnboundLocalError):
        H.control_torque("velocity", g["qpos"][:, 0], g["qvel"][:, 0], g["action"], M_, g["C"], c["jkp"], c["jkd"], c["a_ref"],
                         c["a_scale"], c["torque_lim"], float(g["dt"]))


def test_gae():
    g = load_golden("gae.npz")
    adv, ret, _ = G.estimate_advantages(g["rewards"], g["masks"], g["values"], float(g["gamma"]), float(g["tau"]))
    np.testing.assert_allclose(adv, g["adv"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(ret, g["ret"], rtol=1e-11, atol=1e-12)
    adv, ret, _ = G.estimate_advantages(g["rewards"], g["masks2"], g["values"], float(g["gamma2"]), float(g["tau2"]))
    np.testing.assert_allclose(adv, g["adv2"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(ret, g["ret2"], rtol=1e-11, atol=1e-12)


def test_zfilter_sequence_and_block_merge():
    g = load_golden("zfilter.npz")
    zf = Z.ZFilterOracle(115, clip=5)
    Y = np.stack([zf(x) for x in g["X"]])
    np.testing.assert_allclose(Y, g["Y"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(zf.rs.mean, g["mean"], **TOL)
    np.testing.assert_allclose(zf.rs.std, g["std"], rtol=1e-12, atol=1e-14)
    assert zf.rs.n == int(g["n"])
    np.testing.assert_allclose(np.stack([zf(x, update=False) for x in g["X"][:16]]), g["Yfrozen"], rtol=1e-11, atol=1e-11)
    z1 = Z.ZFilterOracle(115, clip=5)
    np.testing.assert_allclose(z1(g["X"][0]), g["y_first"], rtol=1e-11, atol=1e-11)
    # block (Chan) merge == pushing rows one by one
    rs = Z.RunningStatOracle(115)
    for lo, hi in [(0, 1), (1, 64), (64, 65), (65, 300)]:
        rs.merge_block(g["X"][lo:hi])
    np.testing.assert_allclose(rs.mean, g["mean"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(rs.S, g["S"], rtol=1e-11, atol=1e-10)
    assert rs.n == int(g["n"])


def test_policy_value_nets():
    g = load_golden("policy_value.npz")
    pp, pv = N.params_from_npz(g, "pol_"), N.params_from_npz(g, "val_")
    mean, std = N.policy_mean_std(pp, N.as_t(g["x"]))
    np.testing.assert_allclose(mean.numpy(), g["mean"], **TOL)
    np.testing.assert_allclose(std.numpy(), g["std"], **TOL)
    np.testing.assert_allclose(N.gaussian_log_prob(mean, std, N.as_t(g["a"])).numpy(), g["logp"], rtol=1e-12, atol=1e-11)
    np.testing.assert_allclose(N.value(pv, N.as_t(g["x"])).numpy(), g["value"], **TOL)


def test_video_state_net_modes():
    g = load_golden("video_state_net.npz")
    p = N.params_from_npz(g, "sd_")
    m = int(g["margin"])
    v_out = N.vsnet_test_init(p, g["win"], m)
    np.testing.assert_allclose(v_out.numpy(), g["v_out"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(torch.cat([v_out[[0]], N.as_t(g["st"])], 1).numpy(), g["cat0"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(torch.cat([v_out[[1]], N.as_t(g["st"])], 1).numpy(), g["cat1"], rtol=1e-12, atol=1e-13)
    idx, ctx = N.vsnet_train_ctx(g["masks"], [g["cnn_feat0"], g["cnn_feat1"]], g["v_metas"], m, int(g["cdim"]))
    np.testing.assert_array_equal(idx, g["indices"])
    np.testing.assert_array_equal(ctx, g["cnn_feat_ctx"])
    out = N.vsnet_train_forward(p, ctx, idx, g["states"], m)
    np.testing.assert_allclose(out.numpy(), g["train_out"], rtol=1e-12, atol=1e-13)


def test_ppo_update():
    g = load_golden("ppo_update.npz")
    sdim, adim, cdim, hdim, margin, T_ep = [int(x) for x in g["dims"]]
    P_ = {a: N.params_from_npz(g, "init_%s__" % a) for a in ["p", "p_vs", "v", "v_vs"]}
    batch = {k: g[k] for k in ["states", "actions", "masks", "rewards", "exps", "v_metas"]}
    out = P.update_params(P_["p"], P_["p_vs"], P_["v"], P_["v_vs"], batch, [g["cnn_feat0"], g["cnn_feat1"]],
                          margin=margin, gamma=0.95, tau=0.95, clip_eps=0.2, epochs=3,
                          lr_policy=5e-3, lr_value=3e-3, grad_clip=0.5)
    np.testing.assert_allclose(out["values0"], g["values0"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(out["adv"], g["adv0"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(out["ret"], g["ret0"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(out["logp0"], g["logp0"], rtol=1e-11, atol=1e-11)
    for a in ["p", "p_vs", "v", "v_vs"]:
        for k, v in P_[a].items():
            np.testing.assert_allclose(v.detach().numpy(), g["final_%s__%s" % (a, k)], rtol=1e-9, atol=1e-10,
                                       err_msg="%s.%s" % (a, k))


def test_logger_merge():
    g = load_golden("logger_merge.npz")
    fields = [str(f) for f in g["fields"]]
    logs = []
    for row, ci in zip(g["per_worker"], g["per_worker_c_info"]):
        lg = S.LogOracle()
        for f, v in zip(fields, row):
            setattr(lg, f, v)
        lg.total_c_info = ci
        logs.append(lg)
    mg = S.LogOracle.merge(logs)
    np.testing.assert_allclose([getattr(mg, f) for f in fields], g["merged"], **TOL)
    np.testing.assert_allclose(mg.avg_c_info, g["merged_avg_c_info"], **TOL)


class _ToyEnv:
    def __init__(self):
        self.np_random = np.random.RandomState(5)
        self.t, self.x = 0, None

    def reset(self):
        self.t = 0
        self.x = self.np_random.uniform(-1, 1, size=3)
        return self.x.copy()

    def step(self, a):
        self.t += 1
        self.x = 0.9 * self.x + np.array([a[0], a[1], a[0] - a[1]])
        done = bool(abs(self.x[0]) > 1.5 or self.t >= 7)
        return self.x.copy(), 1.0, done, {"end": self.t >= 7, "fail": abs(self.x[0]) > 1.5}


@pytest.mark.parametrize("use_fork", [False, True])
def test_sampler_semantics_toy(use_fork):
    """Worker quota, never-truncated episodes, pid-ordered concatenation, per-worker RNG de-sync,
    and 'only pid 0 updates the parent's filter' -- Agent.sample (agents/agent.py:29-111)."""
    g = load_golden("sampler_toy.npz")
    p = N.params_from_npz(g, "pol_")

    def select_action(state, t, use_mean):
        mean, std = N.policy_mean_std(p, N.as_t(state).unsqueeze(0), act=torch.tanh)
        a = mean if use_mean else torch.normal(mean, std)
        return a[0].numpy()

    def reward(env, state, action, info):
        return float(np.exp(-np.sum(np.square(action)))), np.array([float(state[0]), float(action[0])])

    rs = Z.ZFilterOracle(3, clip=5)
    torch.manual_seed(33)
    np.random.seed(33)
    batch, log = S.sample(41, 2, _ToyEnv(), select_action, running_state=rs, custom_reward=reward, use_fork=use_fork)
    for k in ["states", "actions", "masks", "next_states", "rewards", "exps"]:
        np.testing.assert_allclose(batch[k], g[k], rtol=1e-12, atol=1e-12, err_msg=k)
    assert log.num_steps == int(g["num_steps"]) and log.num_episodes == int(g["num_episodes"])
    np.testing.assert_allclose(log.avg_c_reward, g["avg_c_reward"], **TOL)
    np.testing.assert_allclose(log.avg_c_info, g["avg_c_info"], **TOL)
    assert rs.rs.n == int(g["rs_n"])
    np.testing.assert_allclose(rs.rs.mean, g["rs_mean"], **TOL)


def test_metrics():
    g = load_golden("metrics.npz")
    dt = float(g["dt"])
    ja = M.joint_angles(g["traj"])
    np.testing.assert_allclose(ja, g["angles"], **TOL)
    jv = M.joint_vels(g["traj"], dt)
    np.testing.assert_allclose(jv, g["vels"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(M.joint_accels(jv, dt), g["accels"], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(M.mean_dist(ja, M.joint_angles(g["traj2"])), g["mean_dist"], **TOL)
    np.testing.assert_allclose(M.mean_abs(g["accels"]), g["mean_abs"], **TOL)


def test_observation_variants():
    """get_full_obs under every combination of cfg.obs_heading / root_deheading / obs_coord / obs_vel (humanoid_v1.py:73-96)."""
    g = load_golden("obs_variants.npz")
    for k, (oh, deheading, root, vel) in enumerate(g["combos"]):
        got = H.full_obs(g["qpos"], g["qvel"], obs_heading=bool(oh), root_deheading=bool(deheading), obs_coord="root" if root else "heading",
                         obs_vel=["full", "root", "no"][vel])
        ref = g["obs_%d" % k]
        assert got.shape == ref.shape == (24, int(oh) + 57 + (58, 6, 0)[vel])
        np.testing.assert_allclose(got, ref, **TOL)


def test_phase_observation():
    """cfg.obs_phase (humanoid_v1.py:92-94): the last column min(cur_t / env_episode_len, 1), cur_t below, at and beyond the length."""
    g = load_golden("obs_phase.npz")
    ph = (g["cur_t"], int(g["episode_len"]))
    got = H.full_obs(g["qpos"], g["qvel"], phase=ph)
    assert got.shape == g["obs_default"].shape == (20, 116)
    np.testing.assert_allclose(got, g["obs_default"], **TOL)
    np.testing.assert_array_equal(got[:, -1], g["obs_default"][:, -1])          # the division itself: bit for bit
    got = H.full_obs(g["qpos"], g["qvel"], obs_heading=True, root_deheading=False, obs_coord="root", obs_vel="root", phase=ph)
    np.testing.assert_allclose(got, g["obs_variant"], **TOL)


def test_random_cur_t_reset_rule():
    """cfg.random_cur_t (humanoid_v1.py:218-220) through the oracle env with a recording physics stand-in: the state set at reset is
    the expert frame start_ind + cur_t, the episode ends after env_episode_len - cur_t steps, step i's expert index is
    start_ind + cur_t, the phase column follows cur_t -- the reference's own reset / step on the same draws (random_cur_t.npz)."""
    import types
    from oracle.cpu_env import OracleHumanoidEnv
    from egopose_amd.skeleton import load_skeleton
    g = load_golden("random_cur_t.npz")
    sk = load_skeleton()
    L, ep_len = int(g["take_len"]), int(g["episode_len"])
    takes = [{"qpos": g["takes_qpos"][k], "qvel": g["takes_qvel"][k], "len": L, "head_height_lb": -100.0} for k in range(g["takes_qpos"].shape[0])]
    cfg = types.SimpleNamespace(fr_margin=int(g["fr_margin"]), env_episode_len=ep_len, random_cur_t=True, obs_phase=True, action_type="torque",
                                jkp=np.ones(52), jkd=np.ones(52), a_ref=np.zeros(52), a_scale=np.ones(52), torque_lim=np.ones(52))

    class Phys:                                   # reset / step / drain of the physics boundary: holds the state it was given
        def reset(self, slot, q, v):
            self.q, self.v = np.array(q), np.array(v)
        def step(self, slot, tau):
            pass
        def drain(self, slot, want_xpos=True):
            xpos = np.zeros((len(sk.body_names), 3)); xpos[:, 2] = 10.0
            return self.q, self.v, np.zeros(sk.nM), np.zeros(sk.nv), xpos

    class Draws:                                  # the oracle's three randint calls return the fixture's draws
        def __init__(self, vals): self.vals = list(vals)
        def randint(self, *a, **k): return self.vals.pop(0)

    for ep in range(len(g["cur_t0"])):
        ph = Phys()
        env = OracleHumanoidEnv(sk, cfg, ph, takes, None)
        env.np_random = Draws([g["expert_ind"][ep], g["start_ind"][ep], g["cur_t0"][ep]])
        ob = env.reset()
        assert (env.expert_ind, env.start_ind, env.cur_t) == (g["expert_ind"][ep], g["start_ind"][ep], g["cur_t0"][ep])
        np.testing.assert_array_equal(ph.q, g["set_qpos"][ep]); np.testing.assert_array_equal(ph.v, g["set_qvel"][ep])
        assert ob[-1] == g["first_phase"][ep]
        n = 0
        while True:
            ob, _, done, info = env.step(np.zeros(52))
            assert env.start_ind + env.cur_t == g["step_index"][ep][n] and ob[-1] == g["step_phase"][ep][n]
            n += 1
            if done:
                assert info["end"] and not info["fail"]
                break
        assert n == g["n_steps"][ep] == ep_len - g["cur_t0"][ep]


def test_constant_and_pose_dist_rewards():
    g = load_golden("reward_simple.npz")
    for i in range(len(g["frame"])):
        r, c = R.constant(bool(g["end"][i]), float(g["end_reward"]))
        assert r == g["constant_reward"][i] == 1.0 and (c == g["constant_cinfo"][i]).all()
        r, c = R.pose_dist(g["qpos"][i], g["expert_qpos"][g["frame"][i]], bool(g["end"][i]), float(g["end_reward"]))
        np.testing.assert_allclose([r, c[0]], [g["pose_dist_reward"][i], g["pose_dist_cinfo"][i][0]], **TOL)
