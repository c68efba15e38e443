"""Product nets (egopose_amd.nets) on CPU float64 against golden outputs of the reference modules:
state_dict keys load strictly (checkpoint drop-in), forward values agree."""
import numpy as np
import torch

from conftest import load_golden
from egopose_amd.nets import MLP, PolicyGaussian, Value, VideoStateNet, RNN


def _sd(g, prefix):
    return {k[len(prefix):]: torch.as_tensor(g[k]) for k in g.files if k.startswith(prefix)}


def test_policy_and_value_match_reference():
    g = load_golden("policy_value.npz")
    torch.set_default_dtype(torch.float64)
    try:
        pol = PolicyGaussian(MLP(13, [10, 6], "relu"), 4, log_std=-2.3, fix_std=True)
        val = Value(MLP(13, [10, 6], "relu"))
        pol.load_state_dict(_sd(g, "pol_"), strict=True)
        val.load_state_dict(_sd(g, "val_"), strict=True)
        assert not pol.action_log_std.requires_grad
        x, a = torch.as_tensor(g["x"]), torch.as_tensor(g["a"])
        with torch.no_grad():
            d = pol(x)
            np.testing.assert_allclose(d.loc.numpy(), g["mean"], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(d.scale.numpy(), g["std"], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(pol.get_log_prob(x, a).numpy(), g["logp"], rtol=1e-12, atol=1e-11)
            np.testing.assert_allclose(val(x).numpy(), g["value"], rtol=1e-12, atol=1e-12)
            np.testing.assert_array_equal(pol.select_action(x, True).numpy(), d.loc.numpy())
    finally:
        torch.set_default_dtype(torch.float32)


def test_video_state_net_test_and_train_modes():
    g = load_golden("video_state_net.npz")
    torch.set_default_dtype(torch.float64)
    try:
        cdim, hdim, m = int(g["cdim"]), int(g["hdim"]), int(g["margin"])
        vs = VideoStateNet(cdim, hdim, m, "lstm", None, False)
        vs.load_state_dict(_sd(g, "sd_"), strict=True)
        vs.set_mode("test")
        with torch.no_grad():
            vs.initialize(torch.as_tensor(g["win"]))
            np.testing.assert_allclose(vs.v_out.numpy(), g["v_out"], rtol=1e-12, atol=1e-13)
            st = torch.as_tensor(g["st"])
            np.testing.assert_allclose(vs(st).numpy(), g["cat0"], rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(vs(st).numpy(), g["cat1"], rtol=1e-12, atol=1e-13)
            # batched windows (what the lockstep rollout feeds): same numbers per column
            win = torch.as_tensor(g["win"])
            vs.initialize(torch.stack([win, win.flip(0)], dim=1))
            np.testing.assert_allclose(vs.v_out[:, 0].numpy(), g["v_out"], rtol=1e-12, atol=1e-13)
        vs.set_mode("train")
        cnn = [g["cnn_feat0"], g["cnn_feat1"]]
        vs.initialize((torch.as_tensor(g["masks"]), cnn, g["v_metas"]))
        np.testing.assert_array_equal(vs.indices, g["indices"])
        np.testing.assert_array_equal(vs.cnn_feat_ctx.numpy(), g["cnn_feat_ctx"])
        with torch.no_grad():
            np.testing.assert_allclose(vs(torch.as_tensor(g["states"])).numpy(), g["train_out"], rtol=1e-12, atol=1e-13)
        # device-table gather builds the same context as the per-episode numpy slicing
        table = torch.as_tensor(np.concatenate(cnn, 0))
        vs.attach_feature_table(table, [0, cnn[0].shape[0]])
        vs.initialize((torch.as_tensor(g["masks"]), cnn, g["v_metas"]))
        np.testing.assert_array_equal(vs.cnn_feat_ctx.numpy(), g["cnn_feat_ctx"])
    finally:
        torch.set_default_dtype(torch.float32)


def test_rnn_float32_generic_path_close_to_float64():
    torch.manual_seed(0)
    rnn = RNN(6, 8, "lstm", bi_dir=True).double()
    x = torch.randn(9, 3, 6, dtype=torch.float64)
    with torch.no_grad():
        ref = rnn(x)
        out32 = rnn.float()(x.float())
    assert out32.shape == (9, 3, 8)
    np.testing.assert_allclose(out32.numpy(), ref.numpy(), atol=2e-6)


def test_forecast_config_and_net_match_reference():
    """ForecastConfig schedules == egoforecast_config.py; VideoForecastNet test / train mode == models/video_forecast_net.py."""
    import os, tempfile, yaml
    from egopose_amd.config import ForecastConfig
    from egopose_amd.nets import VideoForecastNet
    g = load_golden("forecast.npz")
    cwd = os.getcwd()
    d = tempfile.mkdtemp()
    try:
        os.chdir(d)
        os.makedirs("datasets/meta")
        yaml.safe_dump({"train": ["a"], "test": ["b"]}, open("datasets/meta/meta_subject_03.yml", "w"))
        cfg = ForecastConfig("subject_03")
        assert (cfg.fr_margin, cfg.env_episode_len, cfg.end_reward, cfg.policy_s_hdim) == (int(g["fr_margin"]), int(g["env_episode_len"]), bool(g["end_reward"]), int(g["policy_s_hdim"]))
        assert cfg.cfg_dir == "results/egoforecast/subject_03" and cfg.reward_weights["decay"] is True
        np.testing.assert_allclose(cfg.jkp, g["jkp"])
        np.testing.assert_allclose(cfg.a_ref, g["a_ref"])
        for it, nr, ls, lr, init in g["adp"]:
            cfg.update_adaptive_params(int(it))
            np.testing.assert_allclose([cfg.adp_noise_rate, cfg.adp_log_std, cfg.adp_policy_lr, cfg.adp_init_noise], [nr, ls, lr, init], rtol=1e-12)
    finally:
        os.chdir(cwd)
    cdim, sdim, vh, sh, margin = (int(v) for v in g["dims"])
    net = VideoForecastNet(cdim, sdim, vh, margin, "lstm", None, sh, "lstm", False).double()
    net.load_state_dict({k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("sd_")})
    assert net.out_dim == vh + sh
    with torch.no_grad():
        net.set_mode("test")
        net.initialize(torch.as_tensor(g["win"]))
        np.testing.assert_allclose(net.v_out.numpy(), g["v_out"], rtol=1e-10, atol=1e-12)
        outs = np.stack([net(torch.as_tensor(g["st_seq"][k])).numpy()[0] for k in range(g["st_seq"].shape[0])])
        np.testing.assert_allclose(outs, g["test_out"], rtol=1e-10, atol=1e-12)
        # batched test mode + explicit state-net stepping (what the lockstep rollout uses)
        win_b = torch.as_tensor(np.stack([g["win"][:margin], g["win"][:margin] * 0.5], 1))
        ctx = net.context(win_b)
        np.testing.assert_allclose(ctx[0].numpy(), g["v_out"][0], rtol=1e-10, atol=1e-12)
        hc = None
        for k in range(g["st_seq"].shape[0]):
            st = torch.as_tensor(np.repeat(g["st_seq"][k], 2, 0))
            out, hc = net.s_step(st, hc)
            np.testing.assert_allclose(out[0].numpy(), g["test_out"][k][vh:], rtol=1e-10, atol=1e-12)
        net.set_mode("train")
        masks = torch.as_tensor(g["masks"])
        net.initialize((masks, [g["cnn_feat0"], g["cnn_feat1"]], g["v_metas"]))
        np.testing.assert_array_equal(net.indices, g["indices"])
        np.testing.assert_allclose(net(torch.as_tensor(g["states"])).numpy(), g["train_out"], rtol=1e-10, atol=1e-12)
        # gather-built contexts from a device-style table give the same result
        table = torch.as_tensor(np.concatenate([g["cnn_feat0"], g["cnn_feat1"]]))
        net.attach_feature_table(table, [0, g["cnn_feat0"].shape[0]])
        net.initialize((masks, None, g["v_metas"]))
        np.testing.assert_allclose(net(torch.as_tensor(g["states"])).numpy(), g["train_out"], rtol=1e-10, atol=1e-12)


def test_length_bucketed_forward_lstm_is_exact(monkeypatch):
    """Train-mode VideoStateNet with the forward direction run per length bucket == the plain full-window forward:
    outputs and all parameter gradients (float64, CPU)."""
    import egopose_amd.nets as nets
    from egopose_amd.nets import VideoStateNet
    rng = np.random.RandomState(5)
    torch.manual_seed(5)
    cdim, hdim, margin, T_ep = 6, 8, 3, 14
    cnn_feat = [rng.normal(size=(70, cdim)), rng.normal(size=(55, cdim))]
    lens = [14, 2, 9, 14, 1, 7, 3, 3, 11, 5, 14, 6, 2, 8, 1, 4, 12, 10, 2, 13]
    masks, metas = [], []
    for L in lens:
        e = int(rng.randint(2))
        s = int(rng.randint(margin, cnn_feat[e].shape[0] - T_ep - margin))
        masks += [1.0] * (L - 1) + [0.0]
        metas += [[e, s]] * L
    masks, metas = torch.tensor(masks, dtype=torch.float64), np.array(metas)
    states = torch.tensor(rng.normal(size=(len(masks), 5)))
    w = torch.tensor(rng.normal(size=(len(masks), hdim + 5)))
    outs, grads = [], []
    for buckets in (1, 4):
        monkeypatch.setattr(nets, "_FWD_BUCKETS", buckets)
        torch.manual_seed(9)
        vs = VideoStateNet(cdim, hdim, margin, "lstm", None, False).double()
        vs.set_mode("train")
        vs.initialize((masks, cnn_feat, metas))
        assert (vs._buckets is not None) == (buckets > 1)
        out = vs(states)
        (out * w).sum().backward()
        outs.append(out.detach().numpy())
        grads.append({n: p.grad.numpy().copy() for n, p in vs.named_parameters()})
    np.testing.assert_allclose(outs[1], outs[0], rtol=1e-12, atol=1e-13)
    for n in grads[0]:
        np.testing.assert_allclose(grads[1][n], grads[0][n], rtol=1e-10, atol=1e-12, err_msg=n)


def test_windows_that_leave_their_take_are_refused():
    """The device feature table concatenates all takes: a window must not spill into its neighbour (the reference's
    per-take numpy slice comes up short and raises, models/video_state_net.py:52-55)."""
    import pytest
    from egopose_amd.nets import VideoForecastNet, VideoStateNet
    table = torch.zeros(50 + 40, 8)
    net = VideoStateNet(8, 16, 5, "lstm", None, False)
    net.attach_feature_table(table, [0, 50])
    net.check_windows(np.array([0, 1, 1]), np.array([5, 5, 40 - 20 - 5]), 20)           # exactly fits
    with pytest.raises(ValueError, match="leaves take 0"):
        net.check_windows(np.array([0]), np.array([4]), 20)                             # start - margin < 0 (wraps to the table end)
    with pytest.raises(ValueError, match="leaves take 1"):
        net.check_windows(np.array([0, 1]), np.array([10, 16]), 20)                     # runs past the end of take 1
    with pytest.raises(ValueError, match="leaves take 0"):
        net.check_windows(np.array([0]), np.array([26]), 20)                            # would read take 1's first frame
    with pytest.raises(ValueError, match="take index"):
        net.check_windows(np.array([2]), np.array([10]), 20)
    net.set_mode("train")
    masks = torch.tensor([1.0, 1.0, 0.0])
    with pytest.raises(ValueError, match="leaves take"):
        net.initialize((masks, None, np.array([[0, 47]] * 3)))
    fnet = VideoForecastNet(8, 4, 16, 5)
    fnet.attach_feature_table(table, [0, 50])
    fnet.check_windows(np.array([1]), np.array([5]))
    with pytest.raises(ValueError, match="leaves take 1"):
        fnet.check_windows(np.array([1]), np.array([3]))


def test_resnet18_state_dict_is_torchvisions_layout():
    """nets.ResNet18 against the committed key / shape table of torchvision's resnet18 with fc -> 128 (models/resnet.py:6-18;
    tools/make_resnet18_table.py writes it from torchvision's published layout, not from this module): same keys, same order,
    same shapes -- a torchvision checkpoint loads with strict=True -- and the reference's wrapper prefix `resnet.`."""
    import json
    import os
    from egopose_amd.nets import ResNet, ResNet18
    tbl = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "resnet18_keys.json")))
    sd = ResNet18(128).state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == tbl["keys"]
    n = sum(v.numel() for k, v in ResNet18(128).named_parameters() if not k.startswith("fc."))
    assert n == tbl["trainable_backbone_parameters"] == 11176512
    assert list(ResNet(128).state_dict()) == ["resnet." + k for k, _ in tbl["keys"]]
    fake = {k: torch.zeros(shp, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32) for k, shp in tbl["keys"]}
    ResNet18(128).load_state_dict(fake, strict=True)


def test_resnet18_forward_matches_the_functional_restatement():
    """The module's forward (strides, paddings, where the projection shortcuts sit, pooling) against oracle.nets.resnet18_forward,
    an independent torch.nn.functional restatement of the published architecture; float64, eval and train-mode statistics."""
    from egopose_amd.nets import ResNet18
    from oracle.nets import resnet18_forward
    torch.manual_seed(3)
    net = ResNet18(128).double()
    with torch.no_grad():
        for m in net.modules():                      # non-trivial running statistics / affine parameters
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    x = torch.randn(3, 3, 96, 80, dtype=torch.float64)
    sd = net.state_dict()
    with torch.no_grad():
        net.eval()
        np.testing.assert_allclose(net(x).numpy(), resnet18_forward(sd, x).numpy(), rtol=1e-10, atol=1e-10)
        net.train()
        np.testing.assert_allclose(net(x).numpy(), resnet18_forward(sd, x, train=True).numpy(), rtol=1e-9, atol=1e-9)
