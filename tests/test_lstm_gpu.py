"""HIP persistent LSTM (csrc/egp_lstm.hip) vs the oracle's float64 LSTMCell loop (reference: models/rnn.py:45-61):
outputs and parameter gradients, both directions, ragged batch sizes; float32 tolerance 2e-5 relative to scale."""
import numpy as np
import pytest
import torch

from oracle import nets as ON

pytestmark = pytest.mark.gpu


def _oracle(params, x, dy):
    p = {k: v.clone().double().requires_grad_(True) for k, v in params.items()}
    out = ON.bilstm(p, x.double(), prefix="")
    (out * dy.double()).sum().backward()
    return out.detach(), {k: v.grad for k, v in p.items()}


@pytest.mark.parametrize("T,B", [(1, 4), (7, 1), (8, 8), (13, 16), (16, 5), (25, 37), (220, 70)])
def test_hip_lstm_forward_and_gradients(T, B):
    from egopose_amd.nets import RNN
    torch.manual_seed(T * 100 + B)
    rnn = RNN(128, 128, "lstm", bi_dir=True)
    x = torch.randn(T, B, 128)
    dy = torch.randn(T, B, 128)
    ref, gref = _oracle({k: v.detach() for k, v in rnn.state_dict().items()}, x, dy)
    rnn = rnn.cuda()
    xd = x.cuda()
    out = rnn(xd)
    assert out.shape == (T, B, 128)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5)
    (out * dy.cuda()).sum().backward()
    for name, p in rnn.named_parameters():
        g = gref[name].numpy()
        scale = max(1.0, np.abs(g).max())
        np.testing.assert_allclose(p.grad.cpu().numpy() / scale, g / scale, rtol=0, atol=3e-5, err_msg=name)
    with torch.no_grad():                                  # inference path (no saved activations) gives the same output
        out2 = rnn(xd)
    np.testing.assert_allclose(out2.cpu().numpy(), out.detach().cpu().numpy(), rtol=0, atol=1e-6)


def test_hip_lstm_matches_miopen_path(monkeypatch):
    import egopose_amd.nets as nets
    from egopose_amd.nets import RNN
    torch.manual_seed(5)
    rnn = RNN(128, 128, "lstm", bi_dir=True).cuda()
    x = torch.randn(60, 33, 128, device="cuda")
    with torch.no_grad():
        a = rnn(x)
        monkeypatch.setattr(nets, "_LSTM_IMPL", "torch")
        b = rnn(x)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("T,B,D", [(30, 45, 128), (90, 19, 115), (5, 3, 128)])
def test_hip_lstm_hidden128_matches_lstmcell_loop(T, B, D):
    """The hidden-128 instantiation (ego_forecast's causal video net / state net: uni-directional RNN(D, 128)) vs
    torch's float64 nn.LSTMCell loop on the CPU (models/rnn.py:45-61): outputs and all parameter gradients."""
    import egopose_amd.lstm as lstm_mod
    from egopose_amd.nets import RNN
    torch.manual_seed(T + B)
    rnn = RNN(D, 128, "lstm", bi_dir=False)
    x, dy = torch.randn(T, B, D), torch.randn(T, B, 128)
    ref_cell = torch.nn.LSTMCell(D, 128).double()
    ref_cell.load_state_dict({k: v.double() for k, v in rnn.rnn_f.state_dict().items()})
    h = c = torch.zeros(B, 128, dtype=torch.float64)
    outs = []
    for t in range(T):
        h, c = ref_cell(x[t].double(), (h, c))
        outs.append(h)
    ref = torch.stack(outs)
    (ref * dy.double()).sum().backward()
    rnn = rnn.cuda()
    assert lstm_mod.available(x.cuda(), rnn.rnn_f)
    out = rnn(x.cuda())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=3e-5)
    (out * dy.cuda()).sum().backward()
    for name, p in rnn.rnn_f.named_parameters():
        g = getattr(ref_cell, name).grad.numpy()
        scale = max(1.0, np.abs(g).max())
        np.testing.assert_allclose(p.grad.cpu().numpy() / scale, g / scale, rtol=0, atol=5e-5, err_msg=name)


def test_statereg_training_step_on_gpu():
    """VideoRegNet with the ResNet-18 encoder: one fp32 and one bf16-autocast optimisation step on the device (MIOpen
    convolutions + the persistent HIP LSTM), losses finite, every parameter group moves."""
    from egopose_amd.nets import VideoRegNet
    torch.manual_seed(2)
    for ac in (None, torch.bfloat16):
        net = VideoRegNet(115, 128, 128, no_cnn=False, frame_shape=(3, 64, 64)).cuda()
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        x = torch.randn(24, 1, 3, 64, 64, device="cuda")
        gt = torch.randn(24, 115, device="cuda")
        probes = {n: p.detach().clone() for n, p in net.named_parameters() if n in ("cnn.resnet.conv1.weight", "v_net.rnn_f.weight_hh", "linear.weight")}
        if ac is None:
            pred = net(x)
        else:
            with torch.autocast("cuda", dtype=ac):
                pred = net(x).float()
        loss = (gt - pred).pow(2).sum(1).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        assert pred.shape == (24, 115) and np.isfinite(float(loss))
        for n, p in net.named_parameters():
            if n in probes:
                assert not torch.equal(p, probes[n]), n


@pytest.mark.gpu
@pytest.mark.parametrize("T,B,D,H", [(37, 70, 24, 64), (16, 8, 24, 64), (21, 13, 40, 128)])
def test_grouped_sweeps_match_separate_ones(T, B, D, H):
    """lstm.LstmGroup (one grouped launch each way for 4 cells over the same input) == four lstm_direction calls:
    outputs, d_x and every parameter gradient (ragged and full tiles, both hidden sizes)."""
    from egopose_amd import lstm as hl
    dev = torch.device("cuda", 0)
    torch.manual_seed(11)
    cells = [torch.nn.LSTMCell(D, H).to(dev) for _ in range(4)]
    revs = [False, True, False, True]
    x = torch.randn(T, B, D, device=dev)
    ws = [torch.randn(T, B, H, device=dev) for _ in range(4)]
    assert hl.group_available(x, cells)

    def run(grouped):
        for c in cells:
            c.zero_grad()
        xx = x.clone().requires_grad_(True)
        hs = hl.lstm_group(xx, cells, revs) if grouped else [hl.lstm_direction(c, xx, r) for c, r in zip(cells, revs)]
        sum((h * w).sum() for h, w in zip(hs, ws)).backward()
        return [h.detach().clone() for h in hs], xx.grad.clone(), [[p.grad.clone() for p in c.parameters()] for c in cells]

    h_g, dx_g, g_g = run(True)
    h_s, dx_s, g_s = run(False)
    # paired form: the two directions of a cell pair fill the halves of one (T, B, 2H) buffer
    for c in cells:
        c.zero_grad()
    xx = x.clone().requires_grad_(True)
    hp = hl.lstm_group(xx, cells, revs, pairs=True)
    assert len(hp) == 2 and hp[0].shape == (T, B, 2 * H)
    sum((h * torch.cat(ws[2 * i:2 * i + 2], 2)).sum() for i, h in enumerate(hp)).backward()
    for i, h in enumerate(hp):
        torch.testing.assert_close(h.detach(), torch.cat(h_s[2 * i:2 * i + 2], 2), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(xx.grad, dx_s, rtol=1e-4, atol=1e-4)
    for c, gb in zip(cells, g_s):
        for a, b in zip([p.grad for p in c.parameters()], gb):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()) + 1e-6)
    for a, b in zip(h_g, h_s):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dx_g, dx_s, rtol=1e-4, atol=1e-4)
    for ga, gb in zip(g_g, g_s):
        for a, b in zip(ga, gb):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()) + 1e-6)


@pytest.mark.gpu
def test_grouped_video_contexts_match_per_net_contexts():
    """nets.grouped_video_context for (value_vs_net, policy_vs_net) == each net computing its own context."""
    import numpy as np
    from egopose_amd.nets import VideoStateNet, grouped_video_context
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(3)
    cdim, hdim, margin, T_ep = 128, 128, 10, 25
    cnn = [rng.normal(size=(300, cdim))]
    lens = rng.randint(1, T_ep + 1, size=40); lens[0] = T_ep
    masks, metas = [], []
    for Lk in lens:
        s0 = int(rng.randint(margin, cnn[0].shape[0] - T_ep - margin))
        masks += [1.0] * (Lk - 1) + [0.0]
        metas += [[0, s0]] * int(Lk)
    masks = torch.tensor(masks, dtype=torch.float32, device=dev)
    metas = np.array(metas)
    table = torch.tensor(cnn[0], dtype=torch.float32, device=dev)
    states = torch.randn(len(masks), 9, device=dev)
    w = torch.randn(len(masks), hdim + 9, device=dev)
    torch.manual_seed(5)
    nets = [VideoStateNet(cdim, hdim, margin, "lstm", None, False).to(dev) for _ in range(2)]
    res = []
    for grouped in (True, False):
        for n in nets:
            n.zero_grad()
            n.attach_feature_table(table, np.array([0]))
            n.set_mode("train")
            n.initialize((masks, cnn, metas))
        if grouped:
            assert nets[0]._frames is not None             # 300 table rows against 45 x 40 window rows: projection per frame
            assert grouped_video_context(nets)
        ys = [n(states) for n in nets]
        sum((y * w).sum() for y in ys).backward()          # one backward over both nets, as the PPO update does
        res.append([(y.detach().clone(), [p.grad.clone() for p in n.parameters()]) for y, n in zip(ys, nets)])
    for (oa, ga), (ob, gb) in zip(*res):
        torch.testing.assert_close(oa, ob, rtol=1e-5, atol=1e-5)
        for a, b in zip(ga, gb):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()) + 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("dynamic_v", [False, True])
def test_grouped_forecast_contexts_match_per_net_contexts(dynamic_v):
    """nets.grouped_forecast_context for ego_forecast's (value, policy) front ends == each net running its causal video
    LSTM and its state LSTM itself: outputs and every parameter gradient."""
    import numpy as np
    from egopose_amd.nets import VideoForecastNet, grouped_forecast_context
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(4)
    cdim, sdim, hdim, margin, T_ep = 128, 21, 128, 8, 17
    cnn = [rng.normal(size=(200, cdim))]
    lens = rng.randint(1, T_ep + 1, size=23); lens[0] = T_ep
    masks, metas = [], []
    for Lk in lens:
        s0 = int(rng.randint(margin, cnn[0].shape[0] - T_ep - margin))
        masks += [1.0] * (Lk - 1) + [0.0]
        metas += [[0, s0]] * int(Lk)
    masks = torch.tensor(masks, dtype=torch.float32, device=dev)
    metas = np.array(metas)
    table = torch.tensor(cnn[0], dtype=torch.float32, device=dev)
    states = torch.randn(len(masks), sdim, device=dev)
    w = torch.randn(len(masks), 2 * hdim, device=dev)
    torch.manual_seed(6)
    nets = [VideoForecastNet(cdim, sdim, hdim, margin, "lstm", None, hdim, "lstm", dynamic_v).to(dev) for _ in range(2)]
    res = []
    for grouped in (True, False):
        for n in nets:
            n.zero_grad()
            n.attach_feature_table(table, np.array([0]))
            n.set_mode("train")
            n.initialize((masks, cnn, metas))
        if grouped:
            assert grouped_forecast_context(nets, states)
        ys = [n(states) for n in nets]
        assert all(n._grp is None for n in nets)
        sum((y * w).sum() for y in ys).backward()
        res.append([(y.detach().clone(), {k: p.grad.clone() for k, p in n.named_parameters()}) for y, n in zip(ys, nets)])
    for (oa, ga), (ob, gb) in zip(*res):
        torch.testing.assert_close(oa, ob, rtol=1e-5, atol=1e-5)
        for k in gb:
            torch.testing.assert_close(ga[k], gb[k], rtol=1e-4, atol=1e-4 * float(gb[k].abs().max()) + 1e-6, msg=lambda m, k=k: k + ": " + m)


def test_no_grad_passes_take_the_inference_kernels(monkeypatch):
    """torch.no_grad() passes (rollout context pool, values / fixed log-probs) must not save gates and cells: the kernel
    gets NULL save pointers even though the weights require grad (ctx.needs_input_grad ignores the grad mode)."""
    from egopose_amd import lstm

    class Spy:
        def __init__(self, lib):
            self._lib, self.saves = lib, []

        def __getattr__(self, name):
            fn = getattr(self._lib, name)
            if name not in ("egp_lstm_fwd_f32", "egp_lstm_group_fwd_len_f32"):
                return fn
            k = 7 if name == "egp_lstm_fwd_f32" else 9            # gates_out, cells_out follow (include/egopose_hip.h)

            def wrapped(*a):
                self.saves.append((name, a[k].value, a[k + 1].value))
                return fn(*a)
            return wrapped

    spy = Spy(lstm.L.load())
    monkeypatch.setattr(lstm.L, "load", lambda: spy)
    torch.manual_seed(0)
    cells = [torch.nn.LSTMCell(16, 64).cuda() for _ in range(2)]
    x = torch.randn(9, 5, 16, device="cuda")
    with torch.no_grad():
        a = lstm.lstm_direction(cells[0], x, False)
        b = lstm.lstm_group(x, cells, [False, True], pairs=True)[0]
    assert [s[1:] for s in spy.saves] == [(None, None), (None, None)] and not a.requires_grad and not b.requires_grad
    spy.saves.clear()
    a2 = lstm.lstm_direction(cells[0], x, False)
    b2 = lstm.lstm_group(x, cells, [False, True], pairs=True)[0]
    assert all(s[1] is not None and s[2] is not None for s in spy.saves) and a2.requires_grad and b2.requires_grad
    torch.testing.assert_close(a2.detach(), a)
    torch.testing.assert_close(b2.detach(), b)
    (a2.sum() + b2.sum()).backward()
    assert all(c.weight_hh.grad is not None for c in cells)


def test_video_reg_net_with_encoder_matches_float64_cpu_forward():
    """Row f4: VideoRegNet with the ResNet-18 encoder (float32, NHWC, MIOpen convolutions + HIP LSTM + HIP GEMM head is
    not used here: plain MLP) on the device against a float64 CPU forward of the same module, train and eval mode."""
    import copy
    from egopose_amd.nets import VideoRegNet
    torch.manual_seed(5)
    net64 = VideoRegNet(115, 128, 128, no_cnn=False, frame_shape=(3, 64, 64)).double()
    x = torch.randn(12, 1, 3, 64, 64, dtype=torch.float64)
    net = copy.deepcopy(net64).float().cuda().channels_last()
    for train in (True, False):
        net64.train(train); net.train(train)
        with torch.no_grad():
            ref = net64(x)
            got = net(x.float().cuda())
        err = float((got.double().cpu() - ref).norm() / ref.norm())
        assert err < 2e-4, (train, err)
    # batch-norm running statistics moved identically in the train-mode pass
    np.testing.assert_allclose(net.cnn.resnet.bn1.running_mean.cpu().numpy(), net64.cnn.resnet.bn1.running_mean.numpy(), rtol=1e-4, atol=1e-5)


def test_bf16_encoder_keeps_float32_master_weights():
    """BASELINE config 4's arithmetic: bf16 encoder copy (nets.Bf16Shadow) against the float32 encoder, and one
    optimisation step through the master / shadow hand-over."""
    from egopose_amd.nets import VideoRegNet
    torch.manual_seed(6)
    net = VideoRegNet(115, 128, 128, no_cnn=False, frame_shape=(3, 64, 64)).cuda().channels_last()
    x = torch.randn(16, 1, 3, 64, 64, device="cuda")
    gt = torch.randn(16, 115, device="cuda")
    net.eval()
    with torch.no_grad():
        ref = net(x)
    net.bf16_encoder()
    assert len(net._enc16.pairs) == 22 and all(s.dtype == torch.bfloat16 and m.dtype == torch.float32 for m, s in net._enc16.pairs)
    assert net._enc16.shadow.resnet.bn1 is net.cnn.resnet.bn1            # normalisation layers are shared (float32)
    assert all(v.dtype in (torch.float32, torch.int64) for v in net.state_dict().values())          # nothing bf16 in a checkpoint
    # kept outputs (test mode, stored CNN features) never see the bf16 copy: the float32 encoder's result at float32 round-off
    # (two MIOpen float32 calls are not bit-identical; the bf16 copy would be 1e-3 .. 1e-2 away)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    with torch.no_grad():
        assert rel(net(x), ref) < 1e-5
        assert rel(net.get_cnn_feature(x), net.cnn(net._frames(x))) < 1e-5
    # the optimisation step's forward (train mode, autograd on) does: bf16 copy against float32, same batch statistics
    net.train()
    import copy
    net32 = copy.deepcopy(net).bf16_encoder(False)
    got, ref_t = net(x), net32(x)
    assert got.requires_grad and 1e-5 < rel(got, ref_t) < 3e-2
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    loss = (gt - net(x)).pow(2).sum(1).mean()
    opt.zero_grad()
    loss.backward()
    net.encoder_grads_ready()
    assert net.cnn.resnet.conv1.weight.grad is not None and net.cnn.resnet.conv1.weight.grad.dtype == torch.float32
    opt.step()
    net.encoder_stepped()
    for n, p in net.named_parameters():
        assert p.dtype == torch.float32 and not torch.equal(p, before[n]), n
    for m, s in net._enc16.pairs:
        assert torch.equal(s, m.to(torch.bfloat16))


@pytest.mark.parametrize("with_rows", [False, True])
@pytest.mark.parametrize("T,B", [(40, 37), (220, 70), (9, 4), (150, 200)])
def test_ragged_grouped_sweeps_match_the_full_ones_where_it_counts(T, B, with_rows):
    """lstm.ragged_order / egp_lstm_group_*_len_f32: with per-sequence step counts the forward-running direction stops
    early. Outputs inside every sequence's own steps (both directions) and ALL parameter gradients must equal the full
    sweeps' when the loss only looks at those outputs -- which is how the update uses the video context (rows beyond an
    episode are never gathered); the skipped outputs are zeros, or -- when row lists make every later product visit only the
    stepped rows -- left unwritten (the sweeps are HBM-bound: lstm.LstmGroup passes leave_skipped)."""
    from egopose_amd import lstm as lstm_mod
    from egopose_amd.nets import RNN
    torch.manual_seed(T + B)
    rng = np.random.RandomState(T * B)
    steps = rng.randint(0, T + 1, size=B)
    steps[0] = T                                            # one full-length sequence, one empty one
    if B > 1:
        steps[1] = 0
    mask = (torch.arange(T).unsqueeze(1) < torch.as_tensor(steps).unsqueeze(0)).float().unsqueeze(2).cuda()     # (T, B, 1)
    dy = torch.randn(T, B, 128, device="cuda") * mask       # no gradient into rows nobody reads
    x = torch.randn(T, B, 128, device="cuda")
    res = []
    # with_rows: the window length is known, so the input projection and the weight gradients of the forward-running
    # direction visit only the (t, b) rows its workgroups step through (row-index operands of egp_gemm_f32)
    rg = lstm_mod.ragged_order(steps, torch.device("cuda"), T=T if with_rows else None)
    assert (rg.rows is not None) == (with_rows and T * B > 1000)
    for ragged in (None, rg):
        torch.manual_seed(3)
        rnn = RNN(128, 128, "lstm", bi_dir=True).cuda()
        rnn.ragged = ragged
        # what the sweeps leave unwritten must not be read by anything: hand the allocator blocks full of NaN to reuse
        for rows_ in (T * B, (T + 2) * B):
            for width in (128, 512, 1024):
                junk = torch.full((rows_, width), float("nan"), device="cuda")
                del junk
        out = rnn(x)
        out.backward(dy)                                     # (= d/d out of sum(out * dy), without touching unwritten rows)
        res.append((out.detach(), {k: p.grad.clone() for k, p in rnn.named_parameters()}))
    (full, gfull), (rag, grag) = res
    H = 64
    # (not bit for bit: which steps fall into the kernel's unrolled main loop and which into its remainder loop depends on
    #  the step count, and the compiler contracts the cell update differently in the two copies -- one ulp)
    inside = mask.bool().expand_as(rag)
    zero = torch.zeros_like(rag)
    np.testing.assert_allclose(torch.where(inside, rag, zero).cpu().numpy(), torch.where(inside, full, zero).cpu().numpy(), rtol=0, atol=1e-6)
    # beyond its own steps a sequence still rides along to the longest of its workgroup: the full sweep's value, then zeros
    # (or, with row lists, whatever the buffer held)
    fwd_r, fwd_f = rag[:, :, :H], full[:, :, :H]
    if rg.rows is None:
        assert bool((((fwd_r - fwd_f).abs() <= 1e-6) | (fwd_r == 0)).all())
    np.testing.assert_array_equal(rag[:, :, H:].cpu().numpy(), full[:, :, H:].cpu().numpy())        # the reversed direction runs it all
    for k in gfull:
        a, b = grag[k].cpu().numpy(), gfull[k].cpu().numpy()
        scale = max(1.0, np.abs(b).max())
        np.testing.assert_allclose(a / scale, b / scale, rtol=0, atol=5e-6, err_msg=k)      # (other split-K partitions, zeros instead of tiny products)


@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("T,B,F", [(40, 37, 300), (150, 200, 2000), (9, 4, 9)])
def test_frame_table_projection_matches_the_per_window_one(T, B, F, ragged):
    """lstm.LstmGroup `frames`: when x[t, b] = table[base[b] + t] (overlapping windows of consecutive frames, as the video
    context's are) the input projection is computed once per table row and the sweeps fetch row base[b] + t of it. Same
    products, same order within a row -> the same outputs and parameter gradients as the per-window projection."""
    from egopose_amd import lstm as lstm_mod
    torch.manual_seed(T + B + F)
    rng = np.random.RandomState(F)
    table = torch.randn(F, 128, device="cuda")
    base = rng.randint(0, F - T + 1, size=B).astype(np.int32)
    base_t = torch.as_tensor(base, device="cuda")
    x = table[(base_t.long().unsqueeze(0) + torch.arange(T, device="cuda").unsqueeze(1))]        # (T, B, D)
    steps = rng.randint(0, T + 1, size=B)
    steps[0] = T
    rg = lstm_mod.ragged_order(steps, torch.device("cuda"), T=T) if ragged else None
    inside = (torch.arange(T).unsqueeze(1) < torch.as_tensor(steps if ragged else np.full(B, T)).unsqueeze(0)).unsqueeze(2).cuda()
    keep = lambda o: torch.where(inside, o, torch.zeros_like(o))      # (rows beyond a sequence's steps may be left unwritten)
    dys = [keep(torch.randn(T, B, 128, device="cuda")) for _ in range(2)]
    res = []
    for frames in (None, (table, base_t)):
        torch.manual_seed(5)
        cells = [torch.nn.LSTMCell(128, 64).cuda() for _ in range(4)]
        for rows_ in (T * B, (T + 2) * B, F):
            for width in (128, 512, 1024):
                junk = torch.full((rows_, width), float("nan"), device="cuda")
                del junk
        outs = lstm_mod.lstm_group(x, cells, [False, True, False, True], pairs=True, ragged=rg, frames=frames)
        torch.autograd.backward(outs, dys)
        res.append(([keep(o.detach()) for o in outs], [p.grad.clone() for c in cells for p in c.parameters()]))
    (o_d, g_d), (o_f, g_f) = res
    for a, b in zip(o_f, o_d):
        np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    for a, b in zip(g_f, g_d):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        scale = max(1.0, np.abs(b).max())
        np.testing.assert_allclose(a / scale, b / scale, rtol=0, atol=2e-6)       # (split-K partial sums meet in any order)
    # and the no-grad form (the value net's passes, the rollout's context): no gates buffer at all
    with torch.no_grad():
        o_n = lstm_mod.lstm_group(x, cells, [False, True, False, True], pairs=True, ragged=rg, frames=(table, base_t))
    for a, b in zip(o_n, o_f):
        np.testing.assert_allclose(keep(a).cpu().numpy(), b.cpu().numpy(), rtol=0, atol=1e-6)


def test_state_regression_head_on_the_gpu_matches_the_reference():
    """videoreg_head.npz: the reference's VideoRegNet(no_cnn=True) (models/video_reg_net.py:10-59) at the shipped widths -- bi-LSTM
    128 -> 2 x 64, MLP [300, 200], Linear -> 115 -- evaluated in float64 by the reference on a (40, 3, 128) clip. The GPU module
    runs it on the persistent HIP LSTM kernels (hidden 64) and the HIP GEMM head in float32 from the float32-rounded weights:
    2e-5 relative to the output scale (float32 rounding of weights and input, float32 accumulation over 220-term sums)."""
    from conftest import load_golden
    from egopose_amd import lstm as LS
    from egopose_amd.nets import VideoRegNet
    g = load_golden("videoreg_head.npz")
    net = VideoRegNet(115, 128, 128, no_cnn=True, mlp_dim=(300, 200))
    net.load_state_dict({k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("sd_")}, strict=True)
    net = net.cuda().eval()
    x = torch.as_tensor(g["x"], device="cuda")
    assert LS.available(x, net.v_net.rnn_f), "the clip must run on the HIP LSTM kernels"
    with torch.no_grad():
        y = net(x)
    scale = float(np.abs(g["y"]).max())
    assert y.shape == g["y"].shape
    np.testing.assert_allclose(y.double().cpu().numpy(), g["y"], rtol=0, atol=2e-5 * max(1.0, scale))
    # the autograd (training) form of the same module gives the same numbers
    y2 = net.train()(x)
    np.testing.assert_allclose(y2.detach().double().cpu().numpy(), g["y"], rtol=0, atol=2e-5 * max(1.0, scale))
