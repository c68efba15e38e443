"""GPU parity of the assembled PPO update against golden runs of the reference's
`AgentEgo.update_params` (ego_pose/core/agent_ego.py:34-57, agents/agent_ppo.py:16-65, agents/agent_pg.py:19-26) and
`VideoStateNet.initialize('train') / forward('train')` (models/video_state_net.py:40-69):

  * float64 on the device with K5 (GAE) through the C-ABI                    -> final parameters at 1e-9
  * float32 through the path bench.py times: device gather of the episode windows, persistent HIP LSTM recurrences in
    grouped launches, row / episode bucket padding, fused Adam                  -> final parameters at fp32 round-off
  * float64 master modules + float32 shadows (what the unmodified driver gets) -> same tolerance, masters stay float64
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from update_fixture import batch_of, build_agent, check_final

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kctx(skel):
    from egopose_amd.hip import EgpContext
    c = load_golden("config_subject_03.npz")
    ctx = EgpContext(skel, c["jkp"], c["jkd"], c["a_ref"], c["a_scale"], c["torque_lim"], c["b_diffw"])
    yield ctx
    ctx.close()


def _attach_tables(agent, g, dtype):
    """Device-resident feature table + take offsets, as AgentEgo.update_params installs them from the rollout's experts."""
    feats = [np.asarray(g["cnn_feat0"], np.float64), np.asarray(g["cnn_feat1"], np.float64)]
    table = torch.as_tensor(np.concatenate(feats, 0), dtype=dtype, device="cuda")
    off = np.concatenate(([0], np.cumsum([f.shape[0] for f in feats])[:-1]))
    for net in (agent.cn.policy_vs_net, agent.cn.value_vs_net):
        net.attach_feature_table(table, off)


def _spy_gae(agent, kctx):
    """Route K5 through `kctx` and keep what it saw / produced."""
    agent._kernel_ctx = lambda: kctx
    inner = agent._advantages

    def adv_fn(rewards, masks, values):
        adv, ret = inner(rewards, masks, values)
        agent._seen = (adv.double().cpu().numpy(), ret.double().cpu().numpy(), values.double().cpu().numpy())
        return adv, ret
    agent._advantages = adv_fn


@pytest.mark.parametrize("fixture", ["ppo_update.npz", "ppo_update_h128.npz"])
@pytest.mark.parametrize("device_gather", [False, True])
def test_update_params_float64_on_device_matches_reference(kctx, fixture, device_gather, monkeypatch):
    monkeypatch.setenv("EGP_NET_DTYPE", "float64")
    g = load_golden(fixture)
    torch.set_default_dtype(torch.float64)
    try:
        agent, mods = build_agent(g, device="cuda")
        assert agent.shadow is None and agent.cdtype == torch.float64
        if device_gather:
            _attach_tables(agent, g, torch.float64)
        _spy_gae(agent, kctx)
        agent.update_params(batch_of(g))
        a, r, v0 = agent._seen
        np.testing.assert_allclose(v0, g["values0"], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(a, g["adv0"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(r, g["ret0"], rtol=1e-10, atol=1e-11)
        check_final(mods, g, rtol=1e-9, atol=1e-10)
    finally:
        torch.set_default_dtype(torch.float32)


def _count_calls(monkeypatch, module, name):
    calls = []
    inner = getattr(module, name)

    def wrapped(*a, **k):
        calls.append(name)
        return inner(*a, **k)
    monkeypatch.setattr(module, name, wrapped)
    return calls


@pytest.mark.parametrize("fixture", ["ppo_update_h128.npz", "ppo_update_h128_s40.npz"])
@pytest.mark.parametrize("mode", ["float32", "float64-masters"])
@pytest.mark.parametrize("buckets", [False, True])
@pytest.mark.parametrize("gemms", ["hip", "hip-2piece", "torch"])
def test_update_params_float32_hip_path_matches_reference(kctx, mode, buckets, gemms, fixture, monkeypatch):
    """The update bench.py times, pinned to the reference's float64 run of the same batch.
    gemms = "hip": every product through egp_gemm_f32 on the bf16 matrix cores with three-piece operands (float32-class
    products; the default); "hip-2piece": two-piece operands (EGP_GEMM_TERMS=3, ~16 mantissa bits per product);
    "torch": the library's float32 products (EGP_GEMM=torch).
    fixture "..._s40": 40 state columns -- from 32 on, the first MLP layer of both nets gathers [context | state] itself
    (gemm.GatherMlpHead: no concatenated input, no scatter pass); with 24 it takes the two-node form."""
    from egopose_amd import gemm as gemm_mod, gemm_tuning, lstm
    monkeypatch.setenv("EGP_GEMM", "torch" if gemms == "torch" else "hip")
    monkeypatch.setenv("EGP_GEMM_TERMS", "3" if gemms == "hip-2piece" else "6")
    g = load_golden(fixture)
    fused_calls = _count_calls(monkeypatch, gemm_mod, "gather_mlp_head")
    if buckets:     # the bucket padding engages at >= 4 buckets of rows / episodes: shrink the buckets to this batch
        monkeypatch.setattr(gemm_tuning, "ROW_BUCKET", 64)
        monkeypatch.setattr(gemm_tuning, "EPISODE_BUCKET", 8)
        monkeypatch.setitem(gemm_tuning._state, "on", True)
    group_calls = _count_calls(monkeypatch, lstm, "lstm_group")
    masters64 = mode == "float64-masters"
    if masters64:
        torch.set_default_dtype(torch.float64)          # ego_pose/ego_mimic.py:31-32
    try:
        agent, mods = build_agent(g, device="cuda", dtype=torch.float64 if masters64 else torch.float32, fused_adam=not masters64)
        assert agent.cdtype == torch.float32 and (agent.shadow is not None) == masters64
        _attach_tables(agent, g, torch.float32)
        _spy_gae(agent, kctx)
        agent.update_params(batch_of(g))
        torch.cuda.synchronize()
        # 3 epochs (the first one's forward also serves as the value / fixed-log-prob pass), every one through the
        # grouped HIP recurrences (4 sweeps per launch)
        assert len(group_calls) == 3, "the persistent HIP LSTM did not run as expected: %r" % (group_calls,)
        # the fused first layer: both nets, 3 epochs -- exactly when its preconditions hold (three-piece HIP products, >= 32
        # state columns; with episode buckets the context may take the sorted-gather path instead)
        if gemms == "hip" and fixture.endswith("_s40.npz"):
            assert len(fused_calls) == 6 or (buckets and len(fused_calls) == 0), fused_calls
        else:
            assert len(fused_calls) == 0, fused_calls
        a, r, v0 = agent._seen
        np.testing.assert_allclose(v0, g["values0"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(r, g["ret0"], rtol=1e-4, atol=1e-5)          # north_star tolerance
        np.testing.assert_allclose(a, g["adv0"], rtol=1e-4, atol=1e-4)
        if gemms != "hip-2piece":
            # float32 round-off against float64; a handful of elements with a ~0 gradient may take Adam's other +-lr step
            check_final(mods, g, rtol=1e-4, atol=3e-6, max_outliers=8, outlier_atol=1.3e-2)
        else:
            # two-piece products carry 2^-16 instead of 2^-24 per product. Adam divides every gradient element by
            # its own running magnitude, so the elements whose gradient is smaller than that round-off (about 1 % of them
            # here) take a visibly different step: at most 5 % of one step (lr = 2e-3 -> 1e-4 per epoch); everything
            # else stays within the float32 tolerance
            n_par = sum(v.numel() for m in mods.values() for v in m.state_dict().values())
            bad = check_final(mods, g, rtol=1e-4, atol=3e-6, max_outliers=n_par // 50, outlier_atol=3e-4)
            assert bad <= n_par // 50
        want = torch.float64 if masters64 else torch.float32
        assert all(v.dtype == want for m in mods.values() for v in m.state_dict().values())
        if masters64:
            for m, s in agent.shadow.pairs:
                assert torch.equal(s, m.float())
    finally:
        torch.set_default_dtype(torch.float32)


def test_video_state_net_train_mode_device_gather_matches_reference():
    """a14 on the device: segmentation indices, padded context windows gathered from the HBM feature table and the
    gathered output rows == the reference's numpy loop (float64 fixture of tools/gen_golden.py G8)."""
    from egopose_amd.nets import VideoStateNet
    g = load_golden("video_state_net.npz")
    cdim, hdim, margin = int(g["cdim"]), int(g["hdim"]), int(g["margin"])
    net = VideoStateNet(cdim, hdim, margin, "lstm", None, False).double()
    net.load_state_dict({k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("sd_")})
    net.cuda()
    feats = [g["cnn_feat0"], g["cnn_feat1"]]
    table = torch.as_tensor(np.concatenate(feats, 0), device="cuda")
    net.attach_feature_table(table, [0, feats[0].shape[0]])
    net.set_mode("train")
    masks = torch.as_tensor(g["masks"], device="cuda")
    net.initialize((masks, None, g["v_metas"]))              # cnn_feat list not needed: windows come from the table
    np.testing.assert_array_equal(net.indices, g["indices"])
    np.testing.assert_array_equal(net.cnn_feat_ctx.cpu().numpy(), g["cnn_feat_ctx"])
    with torch.no_grad():
        out = net(torch.as_tensor(g["states"], device="cuda"))
    np.testing.assert_allclose(out.cpu().numpy(), g["train_out"], rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("grouped", [False, True])
def test_video_state_net_train_mode_hip_lstm_matches_reference(grouped, monkeypatch):
    """Same row with the float32 persistent recurrences (hidden 64 per direction): the policy's train-mode input of the
    h128 fixture, alone and computed together with a second net in one grouped launch."""
    from egopose_amd import lstm
    from egopose_amd.nets import VideoStateNet, grouped_video_context
    g = load_golden("ppo_update_h128.npz")
    sdim, adim, cdim, hdim, margin, T_ep = [int(x) for x in g["dims"]]
    nets = []
    for name in ("p_vs", "v_vs"):
        net = VideoStateNet(cdim, hdim, margin, "lstm", None, False)
        net.load_state_dict({k[len("init_%s__" % name):]: torch.as_tensor(g[k]) for k in g.files if k.startswith("init_%s__" % name)})
        nets.append(net.cuda())
    feats = [g["cnn_feat0"], g["cnn_feat1"]]
    table = torch.as_tensor(np.concatenate(feats, 0), dtype=torch.float32, device="cuda")
    masks = torch.as_tensor(g["masks"].astype(np.float32), device="cuda")
    calls = _count_calls(monkeypatch, lstm, "lstm_group")
    for net in nets:
        net.attach_feature_table(table, [0, feats[0].shape[0]])
        net.set_mode("train")
        net.initialize((masks, None, g["v_metas"]))
    np.testing.assert_array_equal(nets[0].indices, g["indices"])
    assert tuple(nets[0].cnn_feat_ctx.shape) == tuple(g["ctx_shape"])
    states = torch.as_tensor(g["states"], dtype=torch.float32, device="cuda")
    with torch.no_grad():
        if grouped:
            assert grouped_video_context(nets[::-1])
        out = nets[0](states)
    assert len(calls) == 1            # the bi-LSTM is one grouped launch either way (2 or 4 problems)
    rows = g["policy_in0_rows"]
    np.testing.assert_allclose(out.double().cpu().numpy()[rows], g["policy_in0"], rtol=2e-5, atol=2e-6)


def test_reusing_the_first_forward_pass_changes_nothing(kctx, monkeypatch):
    """agent.reuse_first_pass (default on): the forward that yields the values for GAE and the fixed log-probs is also
    epoch 0's forward. Against the separate no-grad passes of the reference's flow: same parameters to round-off."""
    g = load_golden("ppo_update_h128.npz")
    outs = []
    for reuse in ("1", "0"):
        agent, mods = build_agent(g, device="cuda", dtype=torch.float32, fused_adam=True)
        agent.reuse_first_pass = reuse == "1"
        _attach_tables(agent, g, torch.float32)
        _spy_gae(agent, kctx)
        agent.update_params(batch_of(g))
        outs.append(({k: v.clone() for m in mods.values() for k, v in m.state_dict().items()}, agent._seen))
    (p1, s1), (p0, s0) = outs
    np.testing.assert_allclose(s1[2], s0[2], rtol=1e-6, atol=1e-7)          # values (inference vs training LSTM kernels)
    for k in p1:
        np.testing.assert_allclose(p1[k].cpu().numpy(), p0[k].cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("mode", ["float32", "float64-masters"])
def test_fused_update_tail_equals_the_torch_formulation(kctx, mode, monkeypatch):
    """Round 3's tail -- optim.ppo_losses (one launch for both losses and their gradients w.r.t. values / action mean) and
    optim.FlatUpdater (clip + both Adam steps over flat buffers) -- against the same update with the losses as torch element-wise
    ops under autograd and the caller's own `clip_grad_norm_` + `torch.optim.Adam.step()` (agent.use_fused_loss / use_fused_optim = False):
    same per-epoch losses, same parameters to float32 round-off, and the reference's run is matched by both."""
    from egopose_amd.optim import FlatUpdater
    g = load_golden("ppo_update_h128_s40.npz")
    masters64 = mode == "float64-masters"
    outs = []
    if masters64:
        torch.set_default_dtype(torch.float64)
    try:
        for fused in ("1", "0"):
            agent, mods = build_agent(g, device="cuda", dtype=torch.float64 if masters64 else torch.float32, fused_adam=False)
            agent.use_fused_loss = agent.use_fused_optim = fused == "1"
            _attach_tables(agent, g, torch.float32)
            _spy_gae(agent, kctx)
            agent.update_params(batch_of(g))
            torch.cuda.synchronize()
            assert isinstance(agent._get_updater(), FlatUpdater) == (fused == "1")
            assert agent._fused_losses() == (fused == "1")
            if fused == "1":
                up = agent._get_updater()
                first = [q for q in agent._policy_params() if q.requires_grad][0]      # (parameters() leads with the fixed log-std)
                assert up.steps == [3, 3] and float(agent.optimizer_policy.state[first]["step"]) == 3.0
                if masters64:
                    assert up.mdt == torch.float64 and up.cdt == torch.float32 and up.S is not None
            check_final(mods, g, rtol=1e-4, atol=3e-6, max_outliers=8, outlier_atol=1.3e-2)
            outs.append(({k: v.detach().double().cpu().numpy().copy() for m in mods.values() for k, v in m.state_dict().items()},
                         dict(agent.update_stats)))
    finally:
        torch.set_default_dtype(torch.float32)
    (p1, st1), (p0, st0) = outs
    np.testing.assert_allclose(st1["value_loss"], st0["value_loss"], rtol=2e-5)
    np.testing.assert_allclose(st1["surr_loss"], st0["surr_loss"], rtol=2e-4, atol=1e-7)
    n_bad = 0
    for k in p1:
        bad = ~np.isclose(p1[k], p0[k], rtol=1e-4, atol=3e-6)
        n_bad += int(bad.sum())
        assert np.abs(p1[k] - p0[k]).max() <= 1.3e-2, k              # (an element with a ~0 gradient may take Adam's other step)
    assert n_bad <= 8
