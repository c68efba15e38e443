"""csrc/egp_update.hip on the device: the fused PPO losses against the reference's formulation written with torch autograd in
float64 (agents/agent_ppo.py:58-65, agents/agent_pg.py:19-26, core/distributions.py:6-25), and the flat clip + Adam step against
torch.nn.utils.clip_grad_norm_ + torch.optim.Adam (agents/agent_ppo.py:24-30,53-56)."""
import copy
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference_losses(pred, returns, mean, actions, log_std, adv, fixed, clip, n_val, n_exp, rows):
    """float64 torch autograd over the reference's expressions; returns (v_loss, s_loss, d_pred, d_mean, d_log_std, logp)."""
    pred, mean, log_std = (t.detach().double().clone().requires_grad_(True) for t in (pred, mean, log_std))
    returns, actions, adv = returns.double(), actions.double(), adv.double()
    a = actions if rows is None else actions[rows]
    ad = adv if rows is None else adv[rows]
    std = torch.exp(log_std)
    var = std.pow(2)
    logp = (-(a - mean).pow(2) / (2 * var) - 0.5 * math.log(2 * math.pi) - log_std).sum(1, keepdim=True)      # utils/math.py:14-17
    fx = logp.detach() if fixed is None else fixed.double().reshape(-1, 1)
    ratio = torch.exp(logp - fx)
    surr1 = ratio * ad.reshape(-1, 1)
    surr2 = torch.clamp(ratio, 1.0 - clip, 1.0 + clip) * ad.reshape(-1, 1)
    s_loss = -torch.min(surr1, surr2).sum() / n_exp
    v_loss = (pred - returns).pow(2).sum() / n_val
    (v_loss + s_loss).backward()
    return v_loss.item(), s_loss.item(), pred.grad, mean.grad, log_std.grad, logp.detach()


@pytest.mark.parametrize("n,A,with_rows", [(5000, 52, False), (3001, 52, True), (17, 7, False), (1, 52, False), (70000, 52, True)])
def test_fused_ppo_losses_match_the_autograd_formulation(n, A, with_rows):
    from egopose_amd import optim as O
    g = torch.Generator(device="cuda").manual_seed(n + A)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    log_std = (rnd(1, A) * 0.2 - 1.0)
    rows = None
    if with_rows:
        rows = torch.nonzero(torch.rand(n, device="cuda", generator=g) < 0.7).flatten()
    n_pol = n if rows is None else rows.numel()
    actions = rnd(n, A) * 0.3
    mean = (actions if rows is None else actions[rows]) + rnd(n_pol, A) * torch.exp(log_std)
    pred, returns, adv = rnd(n, 1), rnd(n, 1), rnd(n, 1)
    n_val, n_exp = n + 3.0, n_pol + 5.0                     # "global" counts differ from the local ones
    # pass 1: defines the fixed log-probabilities (ratio == 1 everywhere)
    fixed = torch.empty(n_pol, device="cuda")
    losses, d_pred, d_mean, d_ls = O.ppo_losses(pred, returns, mean, actions, log_std, adv, fixed, True, 0.2, n_val, n_exp, rows=rows,
                                                want_d_log_std=True)
    v, s, gp, gm, gl, logp = _reference_losses(pred, returns, mean, actions, log_std, adv, None, 0.2, n_val, n_exp, rows)
    got = losses.tolist()
    assert abs(got[0] - v) <= 1e-6 * max(1.0, abs(v)) and abs(got[1] - s) <= 2e-6 * max(1.0, abs(s))
    np.testing.assert_allclose(fixed.cpu().numpy(), logp.reshape(-1).cpu().numpy(), rtol=2e-6, atol=2e-5)
    np.testing.assert_allclose(d_pred.cpu().numpy(), gp.cpu().numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(d_mean.cpu().numpy(), gm.cpu().numpy(), rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(d_ls.cpu().numpy(), gl.cpu().numpy(), rtol=2e-4, atol=1e-6)
    # pass 2: the policy has moved -- ratios on both sides of the clip range, both signs of the advantage
    mean2 = mean + rnd(n_pol, A) * 0.02
    losses, d_pred, d_mean, d_ls = O.ppo_losses(pred, returns, mean2, actions, log_std, adv, fixed, False, 0.2, n_val, n_exp, rows=rows,
                                                want_d_log_std=True)
    v, s, gp, gm, gl, logp2 = _reference_losses(pred, returns, mean2, actions, log_std, adv, fixed, 0.2, n_val, n_exp, rows)
    ratio = torch.exp(logp2.reshape(-1) - fixed.double())
    if n_pol > 100:
        assert (ratio > 1.2).any() and (ratio < 0.8).any() and ((ratio > 0.8) & (ratio < 1.2)).any()
    got = losses.tolist()
    assert abs(got[0] - v) <= 1e-6 * max(1.0, abs(v)) and abs(got[1] - s) <= 2e-5 * max(1.0, abs(s))
    # float32 log-probabilities decide which side of the clip boundary a ratio is on: rows within 1e-5 of it may differ
    near = ((ratio - 1.2).abs() < 1e-4) | ((ratio - 0.8).abs() < 1e-4)
    keep = (~near).cpu().numpy()
    np.testing.assert_allclose(d_mean.cpu().numpy()[keep], gm.cpu().numpy()[keep], rtol=2e-4, atol=1e-8)
    assert near.sum().item() <= max(2, n_pol // 200)
    np.testing.assert_allclose(d_ls.cpu().numpy(), gl.cpu().numpy(), rtol=1e-3, atol=1e-5)
    # deterministic: the same call again gives the same bits
    l2, _, m2, ls2 = O.ppo_losses(pred, returns, mean2, actions, log_std, adv, fixed, False, 0.2, n_val, n_exp, rows=rows, want_d_log_std=True)
    assert torch.equal(l2, losses) and torch.equal(m2, d_mean) and torch.equal(ls2, d_ls)


def test_fused_ppo_losses_empty_and_bad_arguments():
    from egopose_amd import optim as O
    z = lambda *s: torch.zeros(*s, device="cuda")
    losses, _, _, _ = O.ppo_losses(z(0, 1), z(0, 1), z(0, 52), z(0, 52), z(1, 52), z(0, 1), z(0), True, 0.2, 1, 1)
    assert losses.tolist() == [0.0, 0.0]
    with pytest.raises(ValueError):
        O.ppo_losses(z(4, 1), z(4, 1), z(4, 52), z(4, 52).double(), z(1, 52), z(4, 1), z(4), True, 0.2, 4, 4)
    with pytest.raises(ValueError):
        O.ppo_losses(z(4, 1), z(4, 1), z(4, 300), z(4, 300), z(1, 300), z(4, 1), z(4), True, 0.2, 4, 4)


def _make_params(dtype, seed):
    torch.manual_seed(seed)
    shapes = [(300, 243), (300,), (200, 300), (200,), (52, 200), (52,), (256, 128), (256, 64), (256,), (1, 52)]
    ps = [torch.nn.Parameter(torch.randn(*s, device="cuda", dtype=dtype) * 0.1) for s in shapes]
    ps[-1].requires_grad_(False)                         # a fixed log-std rides in the optimizer's list without a gradient
    return ps


@pytest.mark.parametrize("master,compute", [(torch.float32, torch.float32), (torch.float64, torch.float32), (torch.float64, torch.float64)])
def test_flat_updater_equals_clip_grad_norm_plus_adam(master, compute):
    """Six steps with a learning rate that changes in between (the driver's set_optimizer_lr), weight decay on one optimizer,
    gradients large enough for the clip to bite on some steps and not on others."""
    from egopose_amd.optim import FlatUpdater
    ref_v, ref_p = _make_params(master, 1), _make_params(master, 2)
    new_v, new_p = copy.deepcopy(ref_v), copy.deepcopy(ref_p)
    mk = lambda v, p: (torch.optim.Adam(v, lr=3e-4, weight_decay=1e-3), torch.optim.Adam(p, lr=5e-5))
    r_ov, r_op = mk(ref_v, ref_p)
    n_ov, n_op = mk(new_v, new_p)
    compute_of = None
    if compute != master:
        compute_of = {m: torch.nn.Parameter(m.detach().to(compute), requires_grad=m.requires_grad) for m in new_v + new_p}
    up = FlatUpdater.build([n_ov, n_op], [(new_p, 40.0)], compute_of)
    assert up is not None and len(up.segments) == 2 and up.segments[1][4] == 1
    comp = lambda m: compute_of[m] if compute_of is not None else m
    g = torch.Generator(device="cuda").manual_seed(5)
    for step in range(6):
        scale = 3.0 if step % 2 else 0.01                   # total norm above / below 40
        if step == 3:
            for o in (r_op, n_op):
                o.param_groups[0]["lr"] = 2e-5
        up.zero_grad()
        for rp, npar in zip(ref_v + ref_p, new_v + new_p):
            if not rp.requires_grad:
                continue
            gr = torch.randn(rp.shape, device="cuda", generator=g, dtype=torch.float32) * scale
            rp.grad = gr.to(master).clone()                   # the reference side sees the same float32 gradient values
            comp(npar).grad = gr.to(compute).clone()          # (own storage: clip_grad_norm_ scales the reference's in place)
        r_ov.step()
        torch.nn.utils.clip_grad_norm_(ref_p, 40.0)
        r_op.step()
        up.collect_grads()
        up.step()
        norm = float(up.norms[1])
        want = math.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in new_p_compute(new_p, comp)))
        assert abs(norm - want) <= 1e-6 * want
        assert (norm > 40.0) == bool(step % 2)
    tol = dict(rtol=1e-12, atol=1e-14) if master == torch.float64 and compute == torch.float64 else \
        (dict(rtol=1e-9, atol=1e-11) if master == torch.float64 else dict(rtol=2e-6, atol=2e-8))
    for rp, npar in zip(ref_v + ref_p, new_v + new_p):
        np.testing.assert_allclose(npar.detach().cpu().numpy(), rp.detach().cpu().numpy(), **tol)
        if compute_of is not None and rp.requires_grad:
            assert torch.equal(compute_of[npar].detach(), npar.detach().to(compute))           # compute copy refreshed by the step
    # the optimizers show the flat moments as their own state; a checkpoint of the modules' tensors still works
    st = n_op.state[new_p[0]]
    assert float(st["step"]) == 6.0 and st["exp_avg"].shape == new_p[0].shape
    np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), r_op.state[ref_p[0]]["exp_avg"].cpu().numpy(), **tol)
    assert n_op.state_dict()["state"][0]["exp_avg_sq"].shape == new_p[0].shape
    assert new_p[-1].requires_grad is False and id(new_p[-1]) not in {id(e[2]) for e in up.entries}


def new_p_compute(params, comp):
    return [comp(p) for p in params if p.requires_grad]


def test_flat_updater_survives_modules_moved_by_the_caller():
    """ego_mimic.py:134-139 wraps checkpoints in `with to_cpu(...)`: module.to() replaces every parameter's storage. The next
    update re-binds the parameters to the flat buffers, moments and step counts intact."""
    from egopose_amd.optim import FlatUpdater
    torch.manual_seed(0)
    net = torch.nn.Linear(16, 8).cuda().double()
    shadow = copy.deepcopy(net).float()
    opt_v = torch.optim.Adam(net.parameters(), lr=1e-2)
    net2 = torch.nn.Linear(4, 4).cuda().double()
    shadow2 = copy.deepcopy(net2).float()
    opt_p = torch.optim.Adam(net2.parameters(), lr=1e-2)
    cof = {m: s for m, s in zip(list(net.parameters()) + list(net2.parameters()), list(shadow.parameters()) + list(shadow2.parameters()))}
    up = FlatUpdater.build([opt_v, opt_p], [], cof)

    def one_step():
        up.zero_grad()
        for s in cof.values():
            s.grad = torch.ones_like(s)
        up.collect_grads()
        up.step()
    one_step()
    before = net.weight.detach().clone()
    net.to("cpu")
    net.to("cuda")                                           # new storage: no longer a view of the flat buffer
    assert net.weight.data_ptr() != up.p_views[0].data_ptr()
    up.rebind()
    assert net.weight.data_ptr() == up.p_views[0].data_ptr() and torch.equal(net.weight.detach(), before)
    one_step()
    assert up.steps == [2, 2] and not torch.equal(net.weight.detach(), before)
    assert torch.equal(shadow.weight.detach(), net.weight.detach().float())


def test_flat_updater_declines_what_it_does_not_implement():
    from egopose_amd.optim import FlatUpdater
    p = [torch.nn.Parameter(torch.zeros(4, device="cuda"))]
    q = [torch.nn.Parameter(torch.zeros(4, device="cuda"))]
    assert FlatUpdater.build([torch.optim.SGD(p, lr=0.1), torch.optim.Adam(q)], []) is None
    assert FlatUpdater.build([torch.optim.Adam(p, amsgrad=True), torch.optim.Adam(q)], []) is None
    cpu = [torch.nn.Parameter(torch.zeros(4))]
    assert FlatUpdater.build([torch.optim.Adam(cpu), torch.optim.Adam(q)], []) is None
    up = FlatUpdater.build([torch.optim.Adam(p), torch.optim.Adam(q)], [])
    up.zero_grad()
    p[0].grad = torch.ones(4, device="cuda")
    with pytest.raises(RuntimeError, match="no gradient"):
        up.collect_grads()                                   # q has none: the fused step would silently move it by its momentum
    up.collect_grads(which=(0,))
    up.step(which=(0,))
    assert up.steps == [1, 0] and float(p[0].detach().abs().sum()) > 0 and float(q[0].detach().abs().sum()) == 0


def test_rccl_single_rank_process_group_runs_the_update_collectives():
    """torch.distributed backend 'nccl' IS RCCL on ROCm. A world of one rank inside the single-GPU lease takes every
    device-side branch of dist.py that the 8-GPU run takes (`_comm_device` == the GPU): the flat gradient all-reduce on the
    updater's own buffer, the moments + counts exchange, the logger and observation-filter merges."""
    import torch.distributed as dist
    from egopose_amd import dist as D
    from egopose_amd.optim import FlatUpdater
    from egopose_amd.rl_core import LoggerRL
    from egopose_amd.zfilter import ZFilter
    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl" and D._comm_device("cuda:0") == torch.device("cuda:0")
        loaded = open("/proc/self/maps").read()
        assert "librccl" in loaded                                              # the collective library really is RCCL
        p = [torch.nn.Parameter(torch.randn(300, 243, device="cuda")), torch.nn.Parameter(torch.randn(300, device="cuda"))]
        q = [torch.nn.Parameter(torch.randn(52, 200, device="cuda"))]
        up = FlatUpdater.build([torch.optim.Adam(p), torch.optim.Adam(q)], [(q, 40.0)])
        up.zero_grad()
        for t in p + q:
            t.grad = torch.randn_like(t)
        up.collect_grads()
        before = up.G.clone()
        up.all_reduce()                                                          # in place on the flat gradient buffer
        torch.cuda.synchronize()
        assert torch.equal(up.G, before)                                         # SUM over one rank
        up.all_reduce(which=(0,))
        up.step()
        fs = D.FlatGradSync(p)
        fs.attach()
        assert p[0].grad.data_ptr() == fs.views[0].data_ptr()
        p[0].grad.add_(1.0)
        fs.all_reduce()
        assert p[0].grad.data_ptr() == fs.views[0].data_ptr() and float(fs.flat[0]) == 1.0
        stats = torch.tensor([1000.0, 0.05, 870.0], dtype=torch.float64, device="cuda")
        out, counts = D.merge_moments_and_counts(stats, (1000, 990))
        assert out.is_cuda and counts == [1000, 990]
        np.testing.assert_allclose(out.cpu().numpy(), [1000.0, 0.05, 870.0], rtol=1e-12)
        assert D.global_count(7, "cuda:0") == 7 and D.global_max(5, "cuda:0") == 5
        lg = LoggerRL.from_totals(10, 2, 10.0, 3, 7, 4.5, 0.1, 0.9, np.arange(5.0))
        merged = D.merge_loggers(lg, "cuda:0")
        assert merged.num_steps == 10 and np.isclose(merged.avg_c_reward, 0.45)
        zf = ZFilter((5,), clip=5)
        X = np.random.RandomState(3).normal(size=(30, 5))
        for x in X[:10]:
            zf(x)
        base = (float(zf.rs._n), zf.rs._M.copy(), zf.rs._S.copy())
        for x in X[10:]:
            zf(x)
        mean, std = zf.rs.mean.copy(), zf.rs.std.copy()
        D.merge_running_state(zf, base, "cuda:0")
        assert zf.rs.n == 30
        np.testing.assert_allclose(zf.rs.mean, mean, rtol=1e-12)
        np.testing.assert_allclose(zf.rs.std, std, rtol=1e-10)
        # the sampling pass's single collective: logger totals + filter deltas in one all-gather, merged on the GPU
        zf2 = ZFilter((5,), clip=5)
        for x in X[:10]:
            zf2(x)
        base2 = (float(zf2.rs._n), zf2.rs._M.copy(), zf2.rs._S.copy())
        for x in X[10:]:
            zf2(x)
        c0 = D.COLLECTIVES["count"]
        merged2 = D.merge_sampling_pass(lg, zf2, base2, "cuda:0")
        assert D.COLLECTIVES["count"] - c0 == 1
        assert merged2.num_steps == 10 and merged2.min_c_reward == 0.1 and merged2.max_c_reward == 0.9
        np.testing.assert_allclose(merged2.avg_c_info, np.arange(5.0) / 10)
        assert zf2.rs.n == 30
        np.testing.assert_allclose(zf2.rs.mean, mean, rtol=1e-12)
        np.testing.assert_allclose(zf2.rs.std, std, rtol=1e-10)
        # Chan merge on the device ignores a NaN mean of an empty contribution
        rows = torch.tensor([[0.0, float("nan"), float("nan")], [4.0, 1.0e9, 2.0]], dtype=torch.float64, device="cuda")
        n_, m_, s_ = D.chan_merge_rows(rows, 1)
        assert float(n_) == 4.0 and float(m_[0]) == 1.0e9 and float(s_[0]) == 2.0
    finally:
        dist.destroy_process_group()
