"""GPU integration: K7 features vs golden expert arrays, the lockstep rollout replayed step by step by the
oracle's one-env CPU env (same physics backend), and one full PPO iteration."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workspace(tmp_path_factory):
    from egopose_amd.bench_support import write_synthetic_dataset
    root = str(tmp_path_factory.mktemp("egp_ws"))
    write_synthetic_dataset(root, "subject_03", n_takes=3, n_frames=300, seed=4)
    return root


def test_pose_features_match_reference_expert_arrays(skel):
    """K7 on the golden take == the arrays the reference's gen_expert formulas produced."""
    from egopose_amd.hip import EgpContext
    c = load_golden("config_subject_03.npz")
    g = load_golden("reward.npz")
    ctx = EgpContext(skel, c["jkp"], c["jkd"], c["a_ref"], c["a_scale"], c["torque_lim"], c["b_diffw"])
    q = g["expert_qpos"]
    ee_w = np.stack([skel.body_xpos(x)[skel.ee_body].ravel() for x in q])
    cur = torch.as_tensor(q, device="cuda")
    prev = torch.cat([cur[:1], cur[:-1]]).contiguous()
    f = {k: v.cpu().numpy() for k, v in ctx.pose_features(cur, prev, torch.as_tensor(ee_w, device="cuda"), expert_convention=True).items()}
    for k in ("rq_rmh", "ee_pos", "bquat"):
        np.testing.assert_allclose(f[k], g["expert_" + k], rtol=1e-10, atol=1e-10, err_msg=k)
    for k in ("qvel", "rlinv_local", "rangv", "bangvel"):       # frame 0 is a copy of frame 1 in the table
        np.testing.assert_allclose(f[k][1:], g["expert_" + k][1:], rtol=1e-9, atol=1e-9, err_msg=k)
    ctx.close()


def _trainer(workspace, n_env, episode_len, **kw):
    from egopose_amd.config import Config
    from egopose_amd.train import Trainer
    os.chdir(workspace)
    cfg = Config("subject_03", create_dirs=False)
    cfg.env_episode_len = episode_len
    cfg.num_optim_epoch = 2
    return Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=n_env, **kw), cfg


@pytest.mark.parametrize("n_groups", [1, 2])
def test_rollout_replayed_by_oracle_env(workspace, skel, n_groups):
    from egopose_amd.physics import SurrogatePhysics
    from oracle.cpu_env import OracleHumanoidEnv
    tr, cfg = _trainer(workspace, 24, 15, num_threads=4, num_groups=n_groups)
    tr.agent.running_state = None                      # raw observations so the replay can compare them
    tr.env.end_reward = 0.37
    batch, log = tr.agent.sample(24 * 20)
    N = len(batch)
    assert log.num_steps == N and N >= 480
    masks = batch.masks
    ends = np.where(masks == 0)[0]
    assert ends[-1] == N - 1, "batch must end on an episode boundary (episodes are never truncated)"
    assert log.num_episodes == len(ends)
    starts = np.r_[0, ends[:-1] + 1]
    lens = ends - starts + 1
    assert lens.max() <= 15 and np.isclose(log.avg_episode_reward, lens.mean())
    np.testing.assert_allclose(log.avg_c_reward, batch.rewards.mean(), rtol=1e-12)
    assert batch.states.dtype == np.float64 and batch.v_metas.dtype == np.int64 and batch.exps.min() == 1
    # replay every episode on the CPU with the oracle env and the recorded actions
    ph = SurrogatePhysics(skel, 1)
    env = OracleHumanoidEnv(skel, cfg, ph, tr.env.expert_arr, tr.env.cnn_feat)
    env.end_reward = 0.37
    for s, e in zip(starts[:12], ends[:12]):
        ei, si = batch.v_metas[s]
        assert (batch.v_metas[s:e + 1] == [ei, si]).all()
        env.expert_ind, env.start_ind, env.cur_t = int(ei), int(si), 0
        ex = tr.env.expert_arr[ei]
        ph.reset(0, ex["qpos"][si], ex["qvel"][si])
        env._drain(True)
        from oracle import humanoid as H
        env.bquat = H.body_quat(env.qpos, skel.body_qpos_start, skel.body_ndof)[0]
        np.testing.assert_allclose(batch.states[s], env._obs(), rtol=1e-9, atol=1e-9)
        for i in range(s, e + 1):
            obs, _, done, info = env.step(batch.actions[i])
            r, _ = env.reward(None, None, info)
            np.testing.assert_allclose(batch.next_states[i], obs, rtol=1e-7, atol=1e-7, err_msg="obs @%d" % i)
            np.testing.assert_allclose(batch.rewards[i], r, rtol=1e-7, atol=1e-7, err_msg="reward @%d" % i)
            assert done == (masks[i] == 0)
            if i < e:
                np.testing.assert_allclose(batch.states[i + 1], obs, rtol=1e-7, atol=1e-7)
    ph.close()
    tr.close()


def test_full_iteration_updates_parameters_and_filter(workspace):
    tr, cfg = _trainer(workspace, 32, 12, num_threads=4, num_groups=2)
    before = [p.detach().clone() for p in tr.policy_net.parameters()]
    log, t_s, t_u, n = tr.iteration(0, 32 * 16)
    assert n >= 512 and np.isfinite(log.avg_c_reward) and 0.0 <= log.min_c_reward <= log.max_c_reward
    assert tr.running_state.rs.n >= n          # every sampled observation (+ resets) went through the filter
    assert np.isfinite(tr.running_state.rs.std).all()
    changed = [not torch.equal(a, b) for a, b in zip(before, tr.policy_net.parameters()) if b.requires_grad]
    assert all(changed)
    assert len(tr.agent.update_stats["surr_loss"]) == 2 and all(np.isfinite(tr.agent.update_stats["value_loss"]))
    # checkpoint round trip in the reference's container (ego_mimic.py:133-139)
    path = os.path.join(workspace, "cp.p")
    tr.save(path)
    from egopose_amd.zfilter import reference_pickle_names
    with reference_pickle_names():           # running_state is pickled under the reference's module path (utils.zfilter)
        cp = pickle.load(open(path, "rb"))
    assert set(cp) == {"policy_dict", "policy_vs_dict", "value_dict", "value_vs_dict", "running_state"}
    assert "v_net.rnn_f.weight_ih" in cp["policy_vs_dict"] and "action_log_std" in cp["policy_dict"]
    tr.load(path)
    # a second iteration reuses the engine and the end_reward bonus
    log2, *_ = tr.iteration(1, 32 * 16)
    assert tr.env.end_reward == pytest.approx(log2.avg_c_reward * cfg.gamma / (1 - cfg.gamma))
    tr.close()


def test_value_front_end_adopts_the_policy_s_train_context(workspace, monkeypatch):
    """update_params initialises the policy's VideoStateNet for the batch and lets the value function's adopt the derived
    context (nets.VideoStateNet.adopt_train_context): every adopted field equals what its own initialize builds, and the
    update's parameters agree with and without sharing (to the 1e-8 that two identical runs differ by: a torch reduction
    in the backward pass is not order-deterministic)."""
    out = {}
    for share in ("1", "0"):
        tr, cfg = _trainer(workspace, 32, 12, num_threads=4, num_groups=2)
        tr.agent.share_train_context = share == "1"
        tr.iteration(0, 32 * 16)
        pv, vv = tr.agent.cn.policy_vs_net, tr.agent.cn.value_vs_net
        if share == "1":
            assert vv.cnn_feat_ctx is pv.cnn_feat_ctx and vv.gather_indices is pv.gather_indices
        else:
            assert vv.cnn_feat_ctx is not pv.cnn_feat_ctx
            assert torch.equal(vv.cnn_feat_ctx, pv.cnn_feat_ctx) and torch.equal(vv._gather_tm, pv._gather_tm)
            assert np.array_equal(vv.indices, pv.indices) and vv._ctx_key == pv._ctx_key
        out[share] = [p.detach().clone() for p in list(tr.value_net.parameters()) + list(tr.value_vs_net.parameters())
                      + list(tr.policy_vs_net.parameters())]
        tr.close()
    for a, b in zip(out["1"], out["0"]):
        torch.testing.assert_close(a, b, rtol=0, atol=1e-6)


def test_update_gradients_hip_path_vs_library_path_at_row_list_scale(workspace, monkeypatch):
    """The update's first backward pass on a rollout batch big enough for everything the small golden fixtures do not reach
    (ragged LSTM sweeps with row lists and unwritten skipped steps, the first MLP layer gathering its own input, split-K
    weight gradients over > 4 096 rows): every parameter gradient of the HIP path against the same pass through the
    library's products and torch's LSTM on the same batch and weights (lr = 0, one epoch: the gradients stay in .grad)."""
    from egopose_amd import gemm as G, lstm as lstm_mod, nets as N
    from egopose_amd.config import Config
    from egopose_amd.train import Trainer
    os.chdir(workspace)
    cfg = Config("subject_03", create_dirs=False)
    cfg.env_episode_len = 150
    cfg.num_optim_epoch = 1
    tr = Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=96, num_threads=4, num_groups=2)
    tr.agent.prefetch_rollout = False          # (its context-pool LSTM sweep would show up in the spies below)
    tr.pre_iter_update(0)
    batch, log = tr.agent.sample(96 * 80)
    for opt in (tr.optimizer_policy, tr.optimizer_value):
        for grp in opt.param_groups:
            grp["lr"] = 0.0
    named = [(n + "." + k, p) for n in ("policy_net", "policy_vs_net", "value_net", "value_vs_net") for k, p in getattr(tr, n).named_parameters()
             if p.requires_grad]
    before = [p.detach().clone() for _, p in named]
    rows_used, fused_used = [], []
    real_group, real_head = lstm_mod.LstmGroup.forward, G.gather_mlp_head

    def run(hip):
        monkeypatch.setenv("EGP_GEMM", "hip" if hip else "torch")
        monkeypatch.setattr(N, "_LSTM_IMPL", "hip" if hip else "torch")
        for _, p in named:
            p.grad = None
        tr.agent.update_params(batch)
        torch.cuda.synchronize()
        return [p.grad.detach().clone() for _, p in named], dict(tr.agent.update_stats)

    # spies: the row lists and the gathering first layer must actually be in play on the HIP pass
    def spy_group(ctx, x, reverse_mask, P, width, train, ragged, *params):
        rows_used.append(ragged is not None and ragged.rows is not None and ragged.rows.shape[0] >= 4096)
        return real_group(ctx, x, reverse_mask, P, width, train, ragged, *params)
    monkeypatch.setattr(lstm_mod.LstmGroup, "forward", staticmethod(spy_group))
    monkeypatch.setattr(G, "gather_mlp_head", lambda *a, **k: (fused_used.append(1), real_head(*a, **k))[1])
    g_hip, st_hip = run(True)
    assert rows_used and all(rows_used), "the batch is too regular for row lists: %r" % (rows_used,)
    assert fused_used, "the gathering first layer did not run"
    g_lib, st_lib = run(False)
    for (name, p), b0, a, b in zip(named, before, g_hip, g_lib):
        assert torch.equal(p.detach(), b0), name                      # lr = 0: nothing moved between the two passes
        scale = max(float(b.abs().max()), 1e-12)
        err = float((a - b).abs().max()) / scale
        assert err < 2e-4, "%s: gradient differs by %.2e of its largest element" % (name, err)
    assert st_hip["value_loss"][0] == pytest.approx(st_lib["value_loss"][0], rel=1e-4)
    assert st_hip["surr_loss"][0] == pytest.approx(st_lib["surr_loss"][0], rel=1e-3, abs=1e-5)
    tr.close()


def test_single_env_facade_matches_oracle_env(workspace, skel):
    """HumanoidEnv.reset/step + reward_func['quat_v3'] on a batch of one == the oracle's CPU env (eval-style use)."""
    from egopose_amd.config import Config
    from egopose_amd.env import HumanoidEnv
    from egopose_amd.physics import SurrogatePhysics
    from egopose_amd.reward import reward_func
    from oracle.cpu_env import OracleHumanoidEnv
    from oracle import humanoid as H
    os.chdir(workspace)
    cfg = Config("subject_03", create_dirs=False)
    cfg.env_episode_len = 6
    env = HumanoidEnv(cfg)
    env.seed(3)
    env.load_experts(cfg.takes["train"], cfg.expert_feat_file, cfg.cnn_feat_file)
    env.end_reward = 0.25
    obs = env.reset()
    ph = SurrogatePhysics(skel, 1)
    ref = OracleHumanoidEnv(skel, cfg, ph, env.expert_arr, env.cnn_feat)
    ref.end_reward = 0.25
    ref.expert_ind, ref.start_ind, ref.cur_t = env.expert_ind, env.start_ind, 0
    ph.reset(0, env.expert["qpos"][env.start_ind], env.expert["qvel"][env.start_ind])
    ref._drain(True)
    ref.bquat = H.body_quat(ref.qpos, skel.body_qpos_start, skel.body_ndof)[0]
    np.testing.assert_allclose(obs, ref._obs(), rtol=1e-10, atol=1e-10)
    assert env.get_episode_cnn_feat().shape == (6 + 2 * cfg.fr_margin, 128)
    rng = np.random.RandomState(0)
    done = False
    while not done:
        a = rng.normal(size=52) * 0.2
        o1, r1, done, info = env.step(a)
        o2, r2, d2, info2 = ref.step(a)
        assert (r1, done, info) == (r2, d2, info2)
        np.testing.assert_allclose(o1, o2, rtol=1e-8, atol=1e-8)
        c1, ci1 = reward_func[cfg.reward_id](env, None, a, info)
        c2, ci2 = ref.reward(None, a, info2)
        np.testing.assert_allclose(c1, c2, rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(ci1, ci2, rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(env.get_ee_pos("heading"), H.ee_pos(ref.qpos, ref.xpos[skel.ee_body].ravel())[0], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(env.data.qpos, ref.qpos, rtol=1e-9, atol=1e-9)
    assert env.cur_t <= 6
    ph.close()
    env.close()


@pytest.mark.parametrize("fail_safe", ["naivefs", "valuefs"])
def test_eval_driver_replayed_by_oracle_env(workspace, skel, fail_safe):
    """Evaluator.eval_expert (ego_mimic_eval.py:103-175 on the single-env facade): every frame of traj_pred, every
    fail-safe re-seat and the reported expert frames are reproduced by the oracle's CPU env driven with the logged
    actions; the metrics of the result pickle are the product-side eval_pose numbers."""
    from egopose_amd.evaluate import Evaluator, compute_metrics
    from egopose_amd import metrics as M
    from egopose_amd.nets import VideoRegNet
    from egopose_amd.physics import SurrogatePhysics
    from oracle.cpu_env import OracleHumanoidEnv
    from oracle import humanoid as H
    tr, cfg = _trainer(workspace, 8, 15, num_threads=2, num_groups=1)
    cfg.env_init_noise = 0.0
    env = tr.env
    m = cfg.fr_margin
    torch.manual_seed(11)
    state_net = VideoRegNet(115, 128, env.cnn_feat[0].shape[-1]).cuda()
    ex = env.expert_arr[0]
    obs_like = np.concatenate([ex["qpos"][m:, 2:], ex["qvel"][m:]], 1)
    mean, std = obs_like.mean(0), np.full(115, 0.02)
    if fail_safe == "valuefs":                               # make the value drop now and then so that resets happen
        with torch.no_grad():
            tr.value_net.value_head.weight.mul_(30.0)
            tr.value_net.value_head.bias.fill_(1.0)
    ev = Evaluator(cfg, env, tr.policy_net, tr.policy_vs_net, tr.value_net, tr.value_vs_net, state_net, mean, std,
                   running_state=tr.running_state, fail_safe=fail_safe, keep_trace=True)
    take = env.expert_list[0]
    results, meta = ev.run(takes=[take])
    pred, orig, vel = results["traj_pred"][take], results["traj_orig"][take], results["vel_pred"][take]
    T = pred.shape[0]
    test_len = env.cnn_feat[0].shape[0] - 2 * m
    assert T == test_len and orig.shape == (T, 59) and vel.shape == (T, 58)
    np.testing.assert_array_equal(orig, ex["qpos"][m:m + T])
    trc = ev.trace[take]
    assert meta == {"algo": "ego_mimic", "num_reset": len(trc["resets"])}
    if fail_safe == "valuefs":
        assert len(trc["resets"]) > 0, "the test is meant to exercise the re-seat path"
    # --- oracle replay
    cfg.env_episode_len = test_len
    ph = SurrogatePhysics(skel, 1)
    ref = OracleHumanoidEnv(skel, cfg, ph, env.expert_arr, env.cnn_feat)
    ref.expert_ind, ref.start_ind, ref.cur_t = 0, m, 0

    def seat(state, ref_qpos):
        qpos = ref_qpos.copy()
        qpos[2:] = state[:57]
        qvel = state[57:].copy()
        M.align_human_state(qpos, qvel, ref_qpos)
        ph.reset(0, qpos, qvel)
        ref._drain(True)
        ref.bquat = H.body_quat(ref.qpos, skel.body_qpos_start, skel.body_ndof)[0]

    seat(trc["state_pred"][0], ex["qpos"][m])
    resets = set(trc["resets"])
    for t in range(T):
        np.testing.assert_allclose(pred[t], ref.qpos, rtol=1e-7, atol=1e-7, err_msg="frame %d" % t)
        np.testing.assert_allclose(vel[t], ref.qvel, rtol=1e-6, atol=1e-6, err_msg="frame %d" % t)
        _, _, _, info = ref.step(trc["actions"][t])
        if info["end"]:
            assert t == T - 1
            break
        if fail_safe == "naivefs":
            assert info["fail"] == (t in resets)
        if t in resets:
            seat(trc["state_pred"][t + 1], ref.qpos)
    ph.close()
    # --- metrics of the result pickle + the reference's file layout
    out = compute_metrics(results)
    assert np.isfinite([out["pose_dist"], out["vel_dist"], out["accels"]]).all() and out["pose_dist"] > 0
    np.testing.assert_allclose(out["per_take"][take][0], M.get_mean_dist(M.get_joint_angles(pred), M.get_joint_angles(orig)))
    cfg.result_dir = os.path.join(workspace, "results_eval")
    path = ev.save(results, meta, 7, data="test")
    assert path.endswith("iter_0007_test%s.p" % ("" if fail_safe == "valuefs" else "_naivefs"))
    r2, m2 = pickle.load(open(path, "rb"))
    assert m2 == meta and set(r2) == {"traj_pred", "traj_orig", "vel_pred"}
    tr.close()


def _forecast_trainer(workspace, n_env, episode_len, **kw):
    from egopose_amd.config import ForecastConfig
    from egopose_amd.train import Trainer
    os.chdir(workspace)
    cfg = ForecastConfig("subject_03", create_dirs=False)
    cfg.env_episode_len = episode_len
    cfg.num_optim_epoch = 2
    return Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=n_env, **kw), cfg


def test_forecast_rollout_matches_train_mode_nets_and_oracle_env(workspace, skel):
    """ego_forecast front end in the lockstep rollout: with the mean action, every recorded action equals the policy
    head over VideoForecastNet's TRAIN-mode forward of the batch (the form pinned to the reference's golden vectors),
    i.e. the per-slot video context and the per-tick state-LSTM stepping (reset at episode starts) are right; physics,
    observations and the decayed reward are replayed by the oracle's CPU env."""
    from egopose_amd.physics import SurrogatePhysics
    from oracle.cpu_env import OracleHumanoidEnv
    from oracle import humanoid as H
    tr, cfg = _forecast_trainer(workspace, 16, 12, num_threads=4, num_groups=2)
    assert tr.forecast and cfg.reward_weights["decay"] and cfg.fr_margin == 30 and not cfg.end_reward
    tr.pre_iter_update(0)
    tr.agent.mean_action = True
    ro = tr.agent._get_rollout()
    ro.mean_action = True
    batch, log = tr.agent.sample(16 * 30)
    N = len(batch)
    assert N >= 480 and batch.exps.max() == 0
    ends = np.where(batch.masks == 0)[0]
    starts = np.r_[0, ends[:-1] + 1]
    assert (ends - starts + 1).max() <= 12
    # --- actions == policy head over the train-mode nets on the same (filtered) states
    dev = torch.device("cuda", 0)
    vs, pol = tr.policy_vs_net, tr.policy_net
    with torch.no_grad():
        vs.set_mode("train")
        vs.attach_feature_table(ro.experts.cnn_table(dev, torch.float32), ro.experts.cnn_offset)
        masks = torch.as_tensor(batch.masks.astype(np.float32), device=dev)
        vs.initialize((masks, tr.env.cnn_feat, batch.v_metas))
        x = vs(torch.as_tensor(batch.states, dtype=torch.float32, device=dev))
        mean, _ = pol.mean_std(x)
        vs.set_mode("test")
    np.testing.assert_allclose(batch.actions, mean.double().cpu().numpy(), rtol=2e-4, atol=2e-4)
    # --- physics / observation / decayed reward replay (raw observations need the filter state: compare rewards + masks)
    ph = SurrogatePhysics(skel, 1)
    env = OracleHumanoidEnv(skel, cfg, ph, tr.env.expert_arr, tr.env.cnn_feat)
    for s, e in zip(starts[:6], ends[:6]):
        ei, si = batch.v_metas[s]
        env.expert_ind, env.start_ind, env.cur_t = int(ei), int(si), 0
        ex = tr.env.expert_arr[ei]
        ph.reset(0, ex["qpos"][si], ex["qvel"][si])
        env._drain(True)
        env.bquat = H.body_quat(env.qpos, skel.body_qpos_start, skel.body_ndof)[0]
        for i in range(s, e + 1):
            _, _, done, info = env.step(batch.actions[i])
            r, _ = env.reward(None, None, info)
            np.testing.assert_allclose(batch.rewards[i], r, rtol=1e-7, atol=1e-7, err_msg="reward @%d" % i)
            assert done == (batch.masks[i] == 0)
    ph.close()
    tr.close()


def test_forecast_rollout_with_phase_observation_and_random_cur_t(workspace, skel):
    """The two ego_forecast env branches (egoforecast_config.py:107-108, humanoid_v1.py:92-94,218-220) together: state width 116
    through the state LSTM, episodes starting inside their window; observations (the phase column included), the decayed reward
    (its t is cur_t) and the episode ends replayed by the oracle env, whose branches are pinned to obs_phase.npz / random_cur_t.npz;
    an update step runs on the batch."""
    from egopose_amd.config import ForecastConfig
    from egopose_amd.train import Trainer
    os.chdir(workspace)
    cfg = ForecastConfig("subject_03", create_dirs=False)
    cfg.env_episode_len = 12
    cfg.num_optim_epoch = 1
    cfg.obs_phase, cfg.random_cur_t = True, True
    tr = Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=16, num_threads=4, num_groups=2)
    assert tr.env.observation_space.shape[0] == 116
    tr.pre_iter_update(0)
    tr.agent.running_state = None
    batch, log = tr.agent.sample(16 * 20)
    ro = tr.agent._get_rollout()
    ends = np.where(batch.masks == 0)[0]
    starts = np.r_[0, ends[:-1] + 1]
    t0 = ro.batch_t0
    np.testing.assert_allclose(batch.states[starts, -1], t0[starts] / 12.0, rtol=0, atol=1e-15)
    assert (ends - starts + 1 <= 12 - t0[starts]).all() and len(set(t0[starts].tolist())) > 4
    _replay_episodes(tr, cfg, skel, batch, range(0, 8), 0.0, t0=t0)
    tr.agent.update_params(batch)
    assert all(np.isfinite(tr.agent.update_stats["value_loss"]))
    tr.close()


def test_forecast_full_iteration(workspace):
    tr, cfg = _forecast_trainer(workspace, 32, 10, num_threads=4, num_groups=2)
    before = [p.detach().clone() for p in list(tr.policy_net.parameters()) + list(tr.policy_vs_net.parameters())]
    log, t_s, t_u, n = tr.iteration(0, 32 * 12)
    assert n >= 384 and np.isfinite(log.avg_c_reward)
    assert tr.env.end_reward == 0.0                     # end_reward: false in the egoforecast configs
    assert cfg.env_init_noise == cfg.adp_init_noise
    after = list(tr.policy_net.parameters()) + list(tr.policy_vs_net.parameters())
    changed = [not torch.equal(a, b) for a, b in zip(before, after) if b.requires_grad]
    assert all(changed), "policy MLP, video LSTM and state LSTM must all receive gradients"
    assert all(np.isfinite(tr.agent.update_stats["value_loss"]))
    path = os.path.join(workspace, "cp_forecast.p")
    tr.save(path)
    from egopose_amd.zfilter import reference_pickle_names
    with reference_pickle_names():
        cp = pickle.load(open(path, "rb"))
    assert "s_net.rnn_f.weight_ih" in cp["policy_vs_dict"] and "v_net.rnn_f.weight_hh" in cp["value_vs_dict"]
    # warm start from an ego_mimic-shaped checkpoint drops the first affine layer (input width differs)
    mim, mcfg = _trainer(workspace, 8, 10, num_threads=2, num_groups=1)
    mpath = os.path.join(workspace, "cp_mimic.p")
    mim.save(mpath)
    w_before = tr.policy_net.net.affine_layers[0].weight.detach().clone()
    tr.warm_start(mpath, mcfg)
    assert torch.equal(tr.policy_net.net.affine_layers[0].weight, w_before)
    assert torch.equal(tr.policy_net.net.affine_layers[1].weight.cpu(), mim.policy_net.net.affine_layers[1].weight.cpu())
    mim.close()
    tr.close()


def _seeded_sample(tr, min_batch, end_reward=0.21):
    """One rollout from a fixed seed state (host + device generators, env sampling stream, a fresh observation filter) on a
    trainer that may have sampled before: everything a bit-equality comparison of two rollouts needs."""
    from egopose_amd.zfilter import ZFilter
    torch.manual_seed(123)
    np.random.seed(5)
    tr.env.seed(77)
    tr.pre_iter_update(0)
    tr.env.end_reward = end_reward
    tr.running_state = tr.agent.running_state = ZFilter((tr.env.observation_space.shape[0],), clip=5)
    ro = tr.agent._get_rollout()
    ro.running_state = tr.running_state
    ro.use_graphs = False                  # (torch tick: eager launches in every run)
    ro.gen.manual_seed(4242)
    ro._pool, ro._pool_pos = None, 0
    torch.cuda.manual_seed(999)
    batch, log = tr.agent.sample(min_batch)
    rs = tr.running_state.rs
    return dict(states=batch.states.copy(), actions=batch.actions.copy(), rewards=batch.rewards.copy(), masks=batch.masks.copy(),
                next_states=batch.next_states.copy(), v_metas=batch.v_metas.copy(), n=rs.n, mean=np.array(rs.mean).copy(),
                std=np.array(rs.std).copy(), steps=log.num_steps, eps=log.num_episodes, r=log.avg_c_reward)


def _assert_same_rollout(a, b, what=""):
    assert a["steps"] == b["steps"] and a["eps"] == b["eps"] and a["n"] == b["n"] and a["r"] == b["r"], what
    for k in ("states", "actions", "rewards", "masks", "next_states", "v_metas", "mean", "std"):
        np.testing.assert_array_equal(a[k], b[k], err_msg="%s %s" % (what, k))


@pytest.mark.parametrize("reward_delay_us", ["0", "500"])
def test_native_tick_is_bit_identical_to_the_torch_tick(workspace, monkeypatch, reward_delay_us):
    """The native tick (pinned flag slabs staged by the policy kernel, two library calls per env-step, the reward riding behind
    the resident K1, the filter's apply pass folded into the next policy step) against the torch-tensor tick with the same
    seeds: identical batches, filter statistics and logger totals. With `reward_delay_us` the reward kernel that rides behind
    the env-step on the engine's stream is held back (EGP_REWARD_JOB_DELAY_US) so that an in-batch reset on the caller's stream
    would overtake it unless egp_engine_reset is ordered behind it: terminal rewards (incl. the end bonus) must still equal the
    torch tick's. Also in the engine's per-substep form (EGP_SERVER=0: the reward is then launched by the tick itself)."""
    outs = {}
    for fast, server in (("0", "1"), ("1", "1"), ("0", "0"), ("1", "0")):
        monkeypatch.setenv("EGP_FAST_TICK", fast)
        monkeypatch.setenv("EGP_SERVER", server)
        monkeypatch.setenv("EGP_REWARD_JOB_DELAY_US", reward_delay_us if fast != "0" else "0")
        tr, cfg = _trainer(workspace, 48, 14, num_threads=4, num_groups=2)
        outs[(fast, server)] = _seeded_sample(tr, 48 * 25)
        assert tr.agent._get_rollout().engine.substeps_per_launch == (15 if server == "1" else 1)
        tr.close()
    # (the two forms of the env-step run different K1 kernels -- one lane per matrix row / a lane grid -- whose sums differ in the
    #  last bits: each tick form is compared within one engine form)
    _assert_same_rollout(outs[("0", "1")], outs[("1", "1")], "resident")
    _assert_same_rollout(outs[("0", "0")], outs[("1", "0")], "per-substep")
    assert abs(outs[("0", "1")]["r"] - outs[("0", "0")]["r"]) < 1e-6


@pytest.mark.parametrize("server", ["1", "0", "ke2"])
def test_rollout_repeats_bit_identically_under_resets_on_every_tick(workspace, monkeypatch, server):
    """Stress of the concurrency the rollout depends on (a cross-stream race was once found by a single
    bit-equality run): 50 rollouts of 96 slots with 6-step episodes -- in-batch resets in nearly every tick of both groups, each
    with its scatter kernel on the caller's stream, the env-step kernel and the reward job on the group's stream, host threads
    re-arming torque rows -- from the same seed state on ONE engine, every one bit-identical to the first; resident and
    per-substep form of the env-step. The reward job is held back 30 us so that orderings that only hold by luck fail."""
    if server == "ke2":                                # a resident wave serving two envs in turn (k_pd_server_tree58_multi)
        monkeypatch.setenv("EGP_SERVER_KE", "2")
    else:
        monkeypatch.setenv("EGP_SERVER", server)
    monkeypatch.setenv("EGP_REWARD_JOB_DELAY_US", "30")
    tr, cfg = _trainer(workspace, 96, 6, num_threads=4, num_groups=2)
    first = _seeded_sample(tr, 96 * 18)
    ro = tr.agent._get_rollout()
    assert ro.engine.substeps_per_launch == (1 if server == "0" else 15)
    assert ro.engine.envs_per_wave == {"1": 1, "0": 0, "ke2": 2}[server]
    ends = np.where(first["masks"] == 0)[0]
    assert len(ends) >= 3 * 96 and ro.timing["ticks"] >= 18          # every slot restarted at least twice
    for rep in range(49):
        _assert_same_rollout(first, _seeded_sample(tr, 96 * 18), "repeat %d" % rep)
    tr.close()


def _train_mode_policy_mean(tr, ro, batch):
    """Policy head over VideoStateNet's TRAIN-mode forward of the batch (the form pinned to the reference's golden
    vectors, tests/test_update_gpu.py): what every recorded mean action must equal."""
    dev = torch.device("cuda", 0)
    vs, pol = tr.agent.cn.policy_vs_net, tr.agent.cn.policy_net
    with torch.no_grad():
        vs.set_mode("train")
        vs.attach_feature_table(ro.experts.cnn_table(dev, torch.float32), ro.experts.cnn_offset)
        masks = torch.as_tensor(batch.masks.astype(np.float32), device=dev)
        vs.initialize((masks, tr.env.cnn_feat, batch.v_metas))
        mean, _ = pol.mean_std(vs(torch.as_tensor(batch.states, dtype=torch.float32, device=dev)))
        vs.set_mode("test")
    return mean.double().cpu().numpy()


@pytest.mark.parametrize("fast_tick", ["1", "0"])
def test_mimic_rollout_actions_equal_policy_of_train_mode_context(workspace, fast_tick, monkeypatch):
    """a1 + a9 + a10 tied together on the ego_mimic path: with the mean action every recorded action equals
    policy(cat(video context of the step's episode and frame, state)). A wrong row of the episode-context pool
    (pre-sampled future episodes), a wrong v_out / t index in the fast tick or a stale context after an in-batch reset
    changes the mean by far more than the tolerance. Both tick implementations; several resets per slot."""
    monkeypatch.setenv("EGP_FAST_TICK", fast_tick)
    tr, cfg = _trainer(workspace, 48, 13, num_threads=4, num_groups=2)
    tr.pre_iter_update(0)
    tr.agent.mean_action = True
    ro = tr.agent._get_rollout()
    ro.pool_batch = 40                      # several refills of the episode pool inside one rollout
    batch, log = tr.agent.sample(48 * 40)
    N = len(batch)
    ends = np.where(batch.masks == 0)[0]
    assert N >= 48 * 40 and len(ends) >= 3 * 48 and batch.exps.max() == 0
    mean = _train_mode_policy_mean(tr, ro, batch)
    np.testing.assert_allclose(batch.actions, mean, rtol=2e-4, atol=2e-4)
    # the check has teeth: shifting the frame index by one moves the mean well outside the tolerance
    sh = batch.v_metas.copy()
    sh[:, 1] += 1
    shifted = type(batch).from_device(**{k: torch.as_tensor(getattr(batch, k)) for k in
                                         ("states", "actions", "masks", "next_states", "rewards", "exps")}, v_metas=torch.as_tensor(sh))
    assert np.abs(_train_mode_policy_mean(tr, ro, shifted) - batch.actions).max() > 1e-3
    tr.close()


def test_sampled_actions_are_policy_mean_plus_unit_noise(workspace):
    """Exploration rollout (fast tick, fused policy kernel, hipGraph noise): (action - mean) / std is N(0, 1) noise, the
    mean being the train-mode policy of the recorded states."""
    tr, cfg = _trainer(workspace, 64, 15, num_threads=4, num_groups=2)
    tr.pre_iter_update(0)
    ro = tr.agent._get_rollout()
    batch, log = tr.agent.sample(64 * 30)
    mean = _train_mode_policy_mean(tr, ro, batch)
    std = float(np.exp(tr.policy_net.action_log_std.detach().cpu().numpy().ravel()[0]))
    z = (batch.actions - mean) / std
    assert batch.exps.min() == 1 and abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02 and np.abs(z).max() < 6.5
    # noise is independent across action dimensions and steps
    assert abs(np.corrcoef(z[:-1, 0], z[1:, 0])[0, 1]) < 0.1 and abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.1
    tr.close()


def _replay_episodes(tr, cfg, skel, batch, episodes, end_reward, tol=1e-7, t0=None, device_dynamics=False):
    """`t0`: cur_t at every batch row's episode start (cfg.random_cur_t; LockstepRollout.batch_t0), default 0.
    `device_dynamics`: the oracle env evaluates M, C itself with the reference's one-substep staleness (OracleHumanoidEnv)."""
    from egopose_amd.physics import SurrogatePhysics
    from oracle.cpu_env import OracleHumanoidEnv
    from oracle import humanoid as H
    ends = np.where(batch.masks == 0)[0]
    starts = np.r_[0, ends[:-1] + 1]
    ph = SurrogatePhysics(skel, 1)
    env = OracleHumanoidEnv(skel, cfg, ph, tr.env.expert_arr, tr.env.cnn_feat, device_dynamics=device_dynamics)
    env.end_reward = end_reward
    for j in episodes:
        s, e = starts[j], ends[j]
        ei, si = batch.v_metas[s]
        assert (batch.v_metas[s:e + 1] == [ei, si]).all()
        c0 = 0 if t0 is None else int(t0[s])
        env.expert_ind, env.start_ind, env.cur_t = int(ei), int(si), c0
        ex = tr.env.expert_arr[ei]
        ph.reset(0, ex["qpos"][si + c0], ex["qvel"][si + c0])
        env.forward()
        env.bquat = H.body_quat(env.qpos, skel.body_qpos_start, skel.body_ndof)[0]
        np.testing.assert_allclose(batch.states[s], env._obs(), rtol=1e-9, atol=1e-9)
        for i in range(s, e + 1):
            obs, _, done, info = env.step(batch.actions[i])
            r, _ = env.reward(None, None, info)
            np.testing.assert_allclose(batch.next_states[i], obs, rtol=tol, atol=tol, err_msg="obs @%d" % i)
            np.testing.assert_allclose(batch.rewards[i], r, rtol=tol, atol=tol, err_msg="reward @%d" % i)
            assert done == (batch.masks[i] == 0)
    ph.close()
    return starts, ends


def test_bench_shape_rollout_replayed_by_oracle_env(tmp_path_factory, skel):
    """BASELINE config 2 exactly as bench.py runs it -- 1 024 env slots, 2 groups, the resident K1 engine, 200-step
    episodes, min batch 50 000, the bench's thread budget -- replayed on a sample of episodes (first / last slots of
    both groups, episodes started by in-batch resets) by the oracle's one-env CPU env."""
    from egopose_amd.bench_support import write_synthetic_dataset
    from egopose_amd.config import Config
    from egopose_amd.physics import default_threads
    from egopose_amd.train import Trainer
    root = str(tmp_path_factory.mktemp("egp_bench_shape"))
    write_synthetic_dataset(root, "subject_03")                     # 8 takes x 2 000 frames, as bench.py
    os.chdir(root)
    cfg = Config("subject_03", create_dirs=False)
    cfg.num_optim_epoch = 1
    n_threads = max(2, default_threads(share=1, device_index=0))
    tr = Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=1024, num_threads=n_threads, num_groups=2)
    tr.pre_iter_update(0)
    tr.agent.running_state = None
    tr.env.end_reward = 1.7
    batch, log = tr.agent.sample(cfg.min_batch_size)
    eng = tr.agent._get_rollout().engine
    assert eng.substeps_per_launch == 15 and eng.n_groups == 2, "bench shape must run the resident K1 engine"
    N = len(batch)
    ends = np.where(batch.masks == 0)[0]
    assert N >= 50000 and ends[-1] == N - 1 and log.num_steps == N
    n_ep = len(ends)
    sample = sorted({0, 1, n_ep // 4, n_ep // 2 - 1, n_ep // 2, n_ep // 2 + 1, n_ep - 2, n_ep - 1})
    starts, _ = _replay_episodes(tr, cfg, skel, batch, sample, 1.7)
    lens = ends - starts + 1
    assert lens.max() <= cfg.env_episode_len and (lens[sample] > 1).all()
    tr.close()


@pytest.mark.parametrize("server", ["1", "0"])
def test_device_dynamics_rollout_replayed_by_oracle_env(workspace, skel, monkeypatch, server):
    """EGP_DEVICE_DYNAMICS=1: M, C of every substep from K8 on the device. The oracle env, which evaluates oracle/dynamics.py with the
    reference's timing (previous substep's state; fresh after a reset), replays the episodes -- in-batch resets included -- in both
    forms of the env-step; and the same episodes do NOT replay with the backend's own (constant) inertia, so the comparison can tell."""
    monkeypatch.setenv("EGP_DEVICE_DYNAMICS", "1")
    monkeypatch.setenv("EGP_SERVER", server)
    tr, cfg = _trainer(workspace, 24, 6, num_threads=4, num_groups=2)
    tr.agent.running_state = None
    tr.env.end_reward = 0.37
    batch, log = tr.agent.sample(24 * 14)
    eng = tr.agent._get_rollout().engine
    assert eng.device_dynamics and eng.substeps_per_launch == (15 if server == "1" else 1)
    ends = np.where(batch.masks == 0)[0]
    n_ep = len(ends)
    assert n_ep >= 48, "every slot must have restarted at least once inside the batch"
    sample = sorted({0, 1, n_ep // 2 - 1, n_ep // 2, n_ep - 2, n_ep - 1})
    _replay_episodes(tr, cfg, skel, batch, sample, 0.37, device_dynamics=True)
    with pytest.raises(AssertionError):
        _replay_episodes(tr, cfg, skel, batch, sample[:1], 0.37, device_dynamics=False)
    tr.close()


def test_2048_slots_take_the_resident_form_and_replay(tmp_path_factory, skel):
    """More slots than the chip holds one-env waves for (2 048 > 4 x CUs): the engine keeps the resident env-step by letting a wave
    serve 2 envs in turn (its own residency probe picks the count) instead of dropping to one launch per substep; a sample of the
    episodes -- first / last slots of both groups, in-batch restarts -- replays on the oracle's CPU env."""
    from egopose_amd.bench_support import write_synthetic_dataset
    from egopose_amd.config import Config
    from egopose_amd.physics import default_threads
    from egopose_amd.train import Trainer
    root = str(tmp_path_factory.mktemp("egp_2048"))
    write_synthetic_dataset(root, "subject_03", n_takes=4, n_frames=600)
    os.chdir(root)
    cfg = Config("subject_03", create_dirs=False)
    cfg.num_optim_epoch = 1
    cfg.env_episode_len = 12
    n_threads = max(2, default_threads(share=1, device_index=0))
    tr = Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=2048, num_threads=n_threads, num_groups=2)
    tr.pre_iter_update(0)
    tr.agent.running_state = None
    tr.env.end_reward = 0.9
    batch, log = tr.agent.sample(2048 * 20)
    eng = tr.agent._get_rollout().engine
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert eng.substeps_per_launch == 15, "2 048 slots must keep the resident env-step"
    assert eng.envs_per_wave >= 2 or cus * 4 >= 2048
    assert 2048 // (4 * eng.envs_per_wave) <= eng.resident_capacity
    ends = np.where(batch.masks == 0)[0]
    n_ep = len(ends)
    assert len(batch) >= 2048 * 20 and ends[-1] == len(batch) - 1
    sample = sorted({0, 1, n_ep // 4, n_ep // 2 - 1, n_ep // 2, n_ep - 2, n_ep - 1})
    _replay_episodes(tr, cfg, skel, batch, sample, 0.9)
    tr.close()


def test_cross_01_config_short_rollout(tmp_path_factory, skel):
    """BASELINE config 3's single-GPU form: the cross_01 config (its own meta / feature ids, 40-take-style dataset)
    loads through Config -> HumanoidEnv.load_experts and a short lockstep rollout is replayed by the oracle env."""
    from egopose_amd.bench_support import write_synthetic_dataset
    from egopose_amd.config import Config
    from egopose_amd.train import Trainer
    root = str(tmp_path_factory.mktemp("egp_cross01"))
    write_synthetic_dataset(root, "cross_01", n_takes=10, n_frames=260, seed=9)
    os.chdir(root)
    cfg = Config("cross_01", create_dirs=False)
    assert cfg.meta_id == "meta_cross_01" and len(cfg.takes["train"]) == 10 and cfg.max_iter_num == 6000
    cfg.env_episode_len = 16
    cfg.num_optim_epoch = 2
    tr = Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=96, num_threads=4, num_groups=2)
    tr.agent.running_state = None
    tr.env.end_reward = 0.4
    batch, log = tr.agent.sample(96 * 24)
    assert set(np.unique(batch.v_metas[:, 0])) <= set(range(10)) and len(np.unique(batch.v_metas[:, 0])) >= 8
    _replay_episodes(tr, cfg, skel, batch, range(0, 12), 0.4)
    log, t_s, t_u, n = tr.iteration(0, 96 * 24)
    assert n >= 96 * 24 and np.isfinite(log.avg_c_reward) and all(np.isfinite(tr.agent.update_stats["value_loss"]))
    tr.close()


@pytest.mark.parametrize("option", [("obs_heading", True), ("obs_vel", "root"), ("root_deheading", False), ("obs_coord", "root"),
                                    ("action_type", "torque"), ("obs_phase", True), ("random_cur_t", True)])
def test_non_default_observation_options_in_the_rollout(workspace, skel, option):
    """The env switches of humanoid_v1.py:73-96,167-172 run through the whole rollout: state width follows the option, the
    recorded observations and rewards are the oracle env's (which evaluates the reference's branches -- obs_coord inside
    the reward too, pinned to the reference by tests/golden/reward_root.npz; the control law by do_simulation.npz), and an
    update step runs."""
    from egopose_amd.config import Config
    from egopose_amd.train import Trainer
    os.chdir(workspace)
    cfg = Config("subject_03", create_dirs=False)
    cfg.env_episode_len = 10
    cfg.num_optim_epoch = 1
    setattr(cfg, *option)
    tr = Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=16, num_threads=2, num_groups=2)
    want = 115 + (1 if option[0] in ("obs_heading", "obs_phase") else 0) - (52 if option[0] == "obs_vel" else 0)
    assert tr.env.observation_space.shape[0] == want and tr.policy_net.net.affine_layers[0].in_features == want + cfg.policy_v_hdim
    tr.agent.running_state = None
    tr.env.end_reward = 0.2
    batch, log = tr.agent.sample(16 * 12)
    assert batch.states.shape[1] == want
    ro = tr.agent._get_rollout()
    if option[0] == "obs_phase":          # (the native tick: the phase column comes from the tick's flag slab)
        assert ro.ctx.obs_phase and ro.timing["ticks"] > 10
        ends = np.where(batch.masks == 0)[0]
        starts = np.r_[0, ends[:-1] + 1]
        assert (batch.states[starts, -1] == 0).all() and np.allclose(batch.next_states[:, -1][ends[ends - starts == 9]], 1.0)
    if option[0] == "random_cur_t":       # episodes start at step cur_t0 of their window and end when cur_t reaches the episode length
        ends = np.where(batch.masks == 0)[0]
        starts = np.r_[0, ends[:-1] + 1]
        t0 = ro.batch_t0
        assert len(set(t0[starts].tolist())) > 3 and t0.max() < 10 and (ends - starts + 1 <= 10 - t0[starts]).all()
        assert (ends - starts + 1 == 10 - t0[starts]).sum() > len(starts) // 2       # (the others fell)
    _replay_episodes(tr, cfg, skel, batch, range(0, 6), 0.2, t0=ro.batch_t0)
    tr.agent.update_params(batch)
    assert all(np.isfinite(tr.agent.update_stats["value_loss"]))
    tr.close()
    with pytest.raises(NotImplementedError):
        cfg.obs_type = "something"
        Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=8, num_threads=2, num_groups=1).agent.sample(8)


@pytest.mark.parametrize("key,value,exc", [("action_type", "velocity", ValueError), ("obs_coord", "world", ValueError),
                                           ("obs_type", "partial", NotImplementedError), ("j_stiff", 5.0, NotImplementedError)])
def test_unsupported_env_options_are_refused(workspace, key, value, exc):
    """Every env key of egomimic_config.py:82-105 / egoforecast_config.py:90-95 is honoured on the HIP path or refused loudly."""
    from egopose_amd.config import Config
    from egopose_amd.train import Trainer
    os.chdir(workspace)
    cfg = Config("subject_03", create_dirs=False)
    cfg.env_episode_len = 10
    setattr(cfg, key, value)
    if key == "j_stiff":
        cfg.action_type = "torque"
    with pytest.raises(exc):
        Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=8, num_threads=2, num_groups=1).agent.sample(8)


def test_global_step_budget_covers_the_batch_without_the_tail(workspace, skel, monkeypatch):
    """EGP_STEP_BUDGET=global: a slot starts a new episode only while collected + in-flight steps fall short of min_batch_size
    (the reference's `while num_steps < min_batch_size` applied to all slots together). The batch still reaches the minimum,
    is never larger than the per-slot rule's, and every recorded step is the oracle env's."""
    from egopose_amd.config import Config
    from egopose_amd.train import Trainer
    os.chdir(workspace)
    sizes = {}
    for mode in ("slot", "global"):
        monkeypatch.setenv("EGP_STEP_BUDGET", mode)
        cfg = Config("subject_03", create_dirs=False)
        cfg.env_episode_len = 12
        tr = Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=32, num_threads=2, num_groups=2)
        tr.agent.running_state = None
        tr.env.end_reward = 0.3
        for min_batch in (32 * 5, 32 * 30):                  # covered by the first episodes / needs restarts
            batch, log = tr.agent.sample(min_batch)
            n_ep = int((batch.masks == 0).sum())
            assert len(batch.masks) >= min_batch and tr.agent._get_rollout().timing["step_budget"] == mode
            sizes[(mode, min_batch)] = (len(batch.masks), n_ep)
            _replay_episodes(tr, cfg, skel, batch, range(0, min(6, n_ep)), 0.3)
        if mode == "global":
            # every episode fails on its first step: the first group parks on the strength of the second group's episodes, which
            # then deliver one step each -- the rollout must still cover the batch (restarts + re-armed slots), in bounded ticks
            tr.env.fix_head_lb = 10.0
            batch, log = tr.agent.sample(32 * 5)
            assert len(batch.masks) >= 32 * 5 and int((batch.masks == 0).sum()) == len(batch.masks)
            assert tr.agent._get_rollout().timing["ticks"] <= 5 + 3 * 12
            tr.env.fix_head_lb = None
        tr.close()
    for min_batch in (32 * 5, 32 * 30):
        assert sizes[("global", min_batch)][0] <= sizes[("slot", min_batch)][0]
    assert sizes[("global", 32 * 5)][1] == 32                # one episode per slot: nothing restarted
    with pytest.raises(ValueError):
        monkeypatch.setenv("EGP_STEP_BUDGET", "bogus")
        tr2 = Trainer(Config("subject_03", create_dirs=False), torch.device("cuda", 0), torch.float32, num_envs=8, num_threads=2, num_groups=1)
        try:
            tr2.agent.sample(8)
        finally:
            tr2.close()


@pytest.mark.parametrize("reward_id", ["pose_dist", "constant"])
def test_rollout_with_the_small_rewards(workspace, skel, reward_id):
    """reward_id 'pose_dist' / 'constant' (reward_function.py:63-80) through the lockstep rollout, replayed by the oracle env."""
    from egopose_amd.config import Config
    from egopose_amd.train import Trainer
    os.chdir(workspace)
    cfg = Config("subject_03", create_dirs=False)
    cfg.env_episode_len = 9
    cfg.num_optim_epoch = 1
    cfg.reward_id = reward_id
    tr = Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=16, num_threads=2, num_groups=2)
    tr.agent.running_state = None
    tr.env.end_reward = 0.6
    batch, log = tr.agent.sample(16 * 12)
    assert np.asarray(log.avg_c_info).shape == (1,)
    if reward_id == "constant":
        assert (batch.rewards == 1.0).all() and log.avg_c_reward == 1.0
    else:
        assert batch.rewards.max() < 5.0 + 0.6 and np.isclose(log.avg_c_info[0], (5.0 - (batch.rewards - 0.6 * (batch.masks == 0))).mean() / 3.0)
    _replay_episodes(tr, cfg, skel, batch, range(0, 6), 0.6)
    tr.close()


def test_custom_reward_none_trains_on_the_env_reward(workspace, skel):
    """agents/agent.py:53-58: with custom_reward=None the memory gets env_reward (HumanoidEnv.step: 1.0 per step,
    humanoid_v1.py:188), the logger c_reward = 0.0 and c_info = [0.0] on every step -- never a silent quat_v3."""
    tr, cfg = _trainer(workspace, 16, 9, num_threads=2, num_groups=2)
    tr.agent.custom_reward = None
    tr.agent._rollout = None
    tr.agent.running_state = None
    tr.env.end_reward = 0.6                       # (would show up in a quat_v3 / pose_dist reward; the env's reward has no bonus)
    batch, log = tr.agent.sample(16 * 12)
    assert tr.agent._get_rollout().reward_kind == "env"
    assert (batch.rewards == 1.0).all()
    assert log.avg_c_reward == 0.0 and log.min_c_reward == 0.0 and log.max_c_reward == 0.0
    np.testing.assert_array_equal(np.asarray(log.avg_c_info), [0.0])
    ends = np.where(batch.masks == 0)[0]
    starts = np.r_[0, ends[:-1] + 1]
    assert log.num_episodes == len(ends) and np.isclose(log.avg_episode_reward, (ends - starts + 1).mean())    # episode_reward = steps x 1.0
    tr.agent.update_params(batch)
    assert all(np.isfinite(tr.agent.update_stats["value_loss"]))
    tr.close()


def test_custom_reward_callable_is_the_reference_plug_point(workspace, skel):
    """agents/agent.py:53-54: `custom_reward(self.env, state, action, info)` may be ANY callable. One without a kernel behind it is
    evaluated on the host per stepped slot through env.SlotView (HumanoidEnv's attribute surface on one slot of the lockstep rollout).
    Two callables written against that surface exactly as the reference's are -- pose_dist_reward (reward_function.py:70-75 with
    env.get_pose_dist) and a quat-space one that reads data.qpos / prev_qpos / bquat / prev_bquat / get_ee_pos / get_expert_attr /
    cur_t / dt / end_reward -- give, from the same seed state, the rewards the registry's KERNELS give (pose_dist: K's arithmetic
    to round-off; quat_v3: the oracle's restatement of reward_function.py:4-60 on the view's fields)."""
    from oracle import reward as R

    def py_pose_dist(env, state, action, info):
        d = env.get_pose_dist()
        r = 5.0 - 3.0 * d
        if info["end"]:
            r += env.end_reward
        return r, np.array([d])

    seen = {"n": 0, "state_dim": None}

    def py_quat_v3(env, state, action, info):
        cfg = env.cfg
        ind = env.get_expert_index(env.cur_t)
        row = {k: env.get_expert_attr(k, ind) for k in ("qpos", "rlinv_local", "rangv", "rq_rmh", "ee_pos", "bquat", "bangvel")}
        seen["n"] += 1
        seen["state_dim"] = np.asarray(state).shape
        assert np.asarray(action).shape == (52,) and np.allclose(env.get_body_quat(), env.bquat)
        r, ci = R.quat_v3(env.data.qpos, env.prev_qpos, env.prev_bquat, env.get_ee_pos(None), env.cur_t, row, cfg.reward_weights, cfg.b_diffw,
                          env.dt, cfg.env_episode_len, info["end"], env.end_reward, env.skel.body_qpos_start, env.skel.body_ndof,
                          obs_coord=getattr(cfg, "obs_coord", "heading"))
        return float(r[0]), ci[0]

    for reward_id, fn, tol in (("pose_dist", py_pose_dist, 1e-12), ("quat_v3", py_quat_v3, 1e-9)):
        out = {}
        for how in ("kernel", "callable"):
            tr, cfg = _trainer(workspace, 16, 9, num_threads=2, num_groups=2)
            cfg.reward_id = reward_id
            from egopose_amd.reward import reward_func
            tr.agent.custom_reward = reward_func[reward_id] if how == "kernel" else fn
            tr.agent._rollout = None
            tr.agent.prefetch_rollout = False
            out[how] = _seeded_sample(tr, 16 * 12, end_reward=0.6)
            assert tr.agent._get_rollout().reward_kind == (reward_id if how == "kernel" else "callable")
            tr.close()
        a, b = out["kernel"], out["callable"]
        np.testing.assert_array_equal(a["states"], b["states"])                 # same trajectories (the reward does not steer a rollout)
        np.testing.assert_array_equal(a["masks"], b["masks"])
        np.testing.assert_allclose(b["rewards"], a["rewards"], rtol=tol, atol=tol, err_msg=reward_id)
        assert a["r"] == pytest.approx(b["r"], rel=1e-9) and (a["masks"] == 0).sum() >= 16
    assert seen["n"] >= 16 * 12 and seen["state_dim"] == (115,)


def test_cross_01_at_1024_slots_replayed_by_oracle_env(tmp_path_factory, skel):
    """BASELINE config 3's per-GPU shard at full size: cross_01 (its own meta / take list, 40 takes) on 1 024 env slots, 2 groups,
    the resident K1 engine, 200-step episodes, min batch 50 000 -- replayed on a sample of episodes by the oracle's CPU env."""
    from egopose_amd.bench_support import write_synthetic_dataset
    from egopose_amd.config import Config
    from egopose_amd.physics import default_threads
    from egopose_amd.train import Trainer
    root = str(tmp_path_factory.mktemp("egp_cross01_full"))
    write_synthetic_dataset(root, "cross_01", n_takes=40, n_frames=420, seed=11)
    os.chdir(root)
    cfg = Config("cross_01", create_dirs=False)
    assert len(cfg.takes["train"]) == 40
    cfg.num_optim_epoch = 1
    n_threads = max(2, default_threads(share=1, device_index=0))
    tr = Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=1024, num_threads=n_threads, num_groups=2)
    tr.pre_iter_update(0)
    tr.agent.running_state = None
    tr.env.end_reward = 0.9
    batch, log = tr.agent.sample(cfg.min_batch_size)
    eng = tr.agent._get_rollout().engine
    assert eng.substeps_per_launch == 15 and eng.n_groups == 2
    N = len(batch)
    ends = np.where(batch.masks == 0)[0]
    assert N >= cfg.min_batch_size and ends[-1] == N - 1 and log.num_steps == N
    assert len(np.unique(batch.v_metas[:, 0])) >= 35                       # the takes of the cross-subject list are all drawn from
    n_ep = len(ends)
    sample = sorted({0, 1, n_ep // 3, n_ep // 2 - 1, n_ep // 2, 2 * n_ep // 3, n_ep - 2, n_ep - 1})
    starts, _ = _replay_episodes(tr, cfg, skel, batch, sample, 0.9)
    assert (ends - starts + 1).max() <= cfg.env_episode_len
    tr.close()


def test_prepared_rollout_setup_changes_nothing(workspace):
    """AgentEgo.update_params sets up the next sampling pass behind its last epoch (LockstepRollout.prepare: record buffers, the
    first reset of every slot, the episode context pool, the noise block) while the GPU finishes the update. Same launches in the
    same stream order: from one seed state, prepare + sample gives the batch sample alone gives, bit for bit (compared without an
    update in between: two runs of the update itself differ in the last bits, a torch reduction of its backward pass is not
    order-deterministic). Through the training loop: the set-up is taken over from the second iteration on, the log_std a driver
    rewrites between update and sample is the one the ticks use, and a set-up made stale by a change of the weights it read or of
    the batch size is dropped and redone."""
    tr, cfg = _trainer(workspace, 64, 12, num_threads=4, num_groups=2)
    tr.agent.prefetch_rollout = False
    plain = _seeded_sample(tr, 64 * 20)

    def prepared_sample(prepared_batch=None):
        ro = tr.agent._get_rollout()
        real = tr.agent.sample

        def sample_with_prepare(min_batch):          # (what update_params does ahead of the driver's next sample call)
            with torch.no_grad():
                ro.noise_rate, ro.mean_action = tr.agent.noise_rate, tr.agent.mean_action
                ro.prepare(prepared_batch or min_batch, end_reward=float(tr.env.end_reward))
            return real(min_batch)
        tr.agent.sample = sample_with_prepare
        try:
            return _seeded_sample(tr, 64 * 20)
        finally:
            tr.agent.sample = real
    ahead = prepared_sample()
    assert tr.agent._get_rollout().timing["setup_prepared"] is True
    _assert_same_rollout(plain, ahead, "prepared")
    # a set-up that is thrown away (made for another batch size) gives back the random streams it drew from: the pass that
    # follows is the pass that would have run without it
    dropped = prepared_sample(prepared_batch=64 * 10)
    assert tr.agent._get_rollout().timing["setup_prepared"] is False
    _assert_same_rollout(plain, dropped, "dropped set-up")
    tr.close()

    tr, cfg = _trainer(workspace, 64, 12, num_threads=4, num_groups=2)
    assert tr.agent.prefetch_rollout
    for it in range(3):
        tr.pre_iter_update(it)
        tr.policy_net.action_log_std.data.fill_(-2.0 - 0.1 * it)            # a log_std schedule (ego_mimic.py:101-102)
        batch, log = tr.agent.sample(64 * 20)
        ro = tr.agent._get_rollout()
        assert ro.timing["setup_prepared"] == (it > 0), it
        assert float(ro._fused.log_std[0]) == pytest.approx(-2.0 - 0.1 * it)
        # the exploration noise really has that scale: action - mean = std * N(0, 1)
        tr.env.end_reward = log.avg_c_reward * cfg.gamma / (1 - cfg.gamma)
        tr.agent.update_params(batch)
        assert ro._prepared is not None and np.isfinite(log.avg_c_reward)
    # the weights the parked set-up read change (a checkpoint load would do this): it must be dropped, not used
    with torch.no_grad():
        tr.policy_net.net.affine_layers[0].weight.mul_(1.0)
    tr.pre_iter_update(3)
    b2, _ = tr.agent.sample(64 * 20)
    assert ro.timing["setup_prepared"] is False and len(b2) >= 64 * 20
    tr.agent.update_params(b2)                            # a different batch size next: dropped as well
    assert ro._prepared is not None
    b3, _ = tr.agent.sample(64 * 10)
    assert ro.timing["setup_prepared"] is False and 64 * 10 <= len(b3) < len(b2)
    tr.close()


def test_forecast_config5_shard_at_1024_slots_replayed_by_oracle_env(tmp_path_factory, skel):
    """BASELINE config 5's per-GPU shard at full size: ego_forecast subject_03 (VideoForecastNet front ends, per-tick state LSTM,
    90-step episodes, decayed reward, no end bonus) on 1 024 env slots, 2 groups, the resident K1 engine, min batch 50 000 --
    physics, termination and the decayed reward of a sample of episodes replayed by the oracle's CPU env, and a PPO iteration on
    the batch."""
    from egopose_amd.bench_support import write_synthetic_dataset
    from egopose_amd.config import ForecastConfig
    from egopose_amd.physics import SurrogatePhysics, default_threads
    from egopose_amd.train import Trainer
    from oracle.cpu_env import OracleHumanoidEnv
    from oracle import humanoid as H
    root = str(tmp_path_factory.mktemp("egp_forecast_full"))
    write_synthetic_dataset(root, "subject_03", n_takes=8, n_frames=600, seed=13)
    os.chdir(root)
    cfg = ForecastConfig("subject_03", create_dirs=False)
    assert cfg.env_episode_len == 90 and cfg.fr_margin == 30 and cfg.policy_s_net == "lstm" and not cfg.end_reward
    cfg.num_optim_epoch = 1
    n_threads = max(2, default_threads(share=1, device_index=0))
    tr = Trainer(cfg, torch.device("cuda", 0), torch.float32, num_envs=1024, num_threads=n_threads, num_groups=2)
    assert tr.forecast
    tr.pre_iter_update(0)
    batch, log = tr.agent.sample(cfg.min_batch_size)
    eng = tr.agent._get_rollout().engine
    assert eng.substeps_per_launch == 15 and eng.n_groups == 2
    N = len(batch)
    ends = np.where(batch.masks == 0)[0]
    starts = np.r_[0, ends[:-1] + 1]
    assert N >= cfg.min_batch_size and ends[-1] == N - 1 and (ends - starts + 1).max() <= 90 and log.num_steps == N
    ph = SurrogatePhysics(skel, 1)
    env = OracleHumanoidEnv(skel, cfg, ph, tr.env.expert_arr, tr.env.cnn_feat)
    n_ep = len(ends)
    for k in sorted({0, 1, n_ep // 3, n_ep // 2, 2 * n_ep // 3, n_ep - 2, n_ep - 1}):
        s, e = starts[k], ends[k]
        ei, si = batch.v_metas[s]
        assert (batch.v_metas[s:e + 1] == [ei, si]).all()
        env.expert_ind, env.start_ind, env.cur_t = int(ei), int(si), 0
        ex = tr.env.expert_arr[ei]
        ph.reset(0, ex["qpos"][si], ex["qvel"][si])
        env._drain(True)
        env.bquat = H.body_quat(env.qpos, skel.body_qpos_start, skel.body_ndof)[0]
        for i in range(s, e + 1):
            _, _, done, info = env.step(batch.actions[i])
            r, _ = env.reward(None, None, info)
            np.testing.assert_allclose(batch.rewards[i], r, rtol=1e-7, atol=1e-7, err_msg="episode %d reward @%d" % (k, i))
            assert done == (batch.masks[i] == 0)
    ph.close()
    tr.agent.update_params(batch)
    assert all(np.isfinite(tr.agent.update_stats["value_loss"])) and all(np.isfinite(tr.agent.update_stats["surr_loss"]))
    tr.close()
