"""The chained MLP launches of the update (csrc/egp_chain.hip, egopose_amd/chain.py) against a float64 torch evaluation of
head(relu-MLP([ctx[idx] | x])) (models/mlp.py:22-25, core/policy_gaussian.py:19-24, core/critic.py:15-18) and against the
layer-per-launch path they replace."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nets(n_out, seed):
    from egopose_amd.nets import MLP
    torch.manual_seed(seed)
    mlp = MLP(243, (300, 200), "relu").cuda()
    head = torch.nn.Linear(200, n_out).cuda()
    with torch.no_grad():
        head.weight.mul_(3.0)
        for l in mlp.affine_layers:
            l.bias.normal_(std=0.3)
    return mlp, head


def _ref64(ctx2d, idx, x, mlp, head, dout):
    c = ctx2d.double().detach().requires_grad_(True)
    ps = [p.double().detach().requires_grad_(True) for l in list(mlp.affine_layers) + [head] for p in (l.weight, l.bias)]
    h = torch.cat((c[idx], x.double()), 1)
    for i in range(2):
        h = torch.relu(h @ ps[2 * i].t() + ps[2 * i + 1])
    out = h @ ps[4].t() + ps[5]
    out.backward(dout.double())
    return out.detach(), c.grad, [p.grad for p in ps]


@pytest.mark.parametrize("n_out,n", [(52, 1111), (1, 640), (52, 37)])
def test_chained_mlp_matches_float64_and_the_layer_per_launch_path(n_out, n, monkeypatch):
    monkeypatch.setenv("EGP_MLP_CHAIN", "1")          # (opt-in: slower than the layer-per-launch path so far, chain.py)
    from egopose_amd import chain as CH
    from egopose_amd import gemm as G
    mlp, head = _nets(n_out, 5 + n_out)
    g = torch.Generator(device="cuda").manual_seed(n)
    R = n + 300
    ctx2d = torch.randn(R, 128, device="cuda", generator=g)
    idx = torch.randperm(R, device="cuda", generator=g)[:n].contiguous()
    x = torch.randn(n, 115, device="cuda", generator=g) * 1.5
    dout = torch.randn(n, n_out, device="cuda", generator=g)
    gi = G.GatheredInput(ctx2d.clone().requires_grad_(True), idx, x)
    assert CH.available(gi, mlp.affine_layers, head)
    out = CH.chain_mlp_head(gi, mlp.affine_layers, head)
    out.backward(dout)
    got_ctx = gi.ctx2d.grad.clone()
    params = [p for l in list(mlp.affine_layers) + [head] for p in (l.weight, l.bias)]
    got_p = [p.grad.clone() for p in params]
    ref_out, ref_ctx, ref_p = _ref64(ctx2d, idx, x, mlp, head, dout)

    def close(a, b, tol, what):
        scale = max(1.0, float(b.abs().max()))
        err = float((a.double() - b).abs().max())
        assert err <= tol * scale, "%s: max error %.3g at scale %.3g" % (what, err, scale)
    close(out.detach(), ref_out, 3e-6, "output")
    close(got_ctx, ref_ctx, 3e-6, "d ctx")
    assert float(got_ctx[torch.ones(R, dtype=torch.bool, device="cuda").index_fill_(0, idx, False)].abs().max()) == 0.0      # rows nobody gathered
    for k, (a, b) in enumerate(zip(got_p, ref_p)):
        close(a, b, 2e-5, "parameter gradient %d" % k)            # sums over n rows in float32
    # the layer-per-launch path (same six-term products, other summation order)
    for p in params:
        p.grad = None
    gi2 = G.GatheredInput(ctx2d.clone().requires_grad_(True), idx, x)
    out2 = G.gather_mlp_head_layers(gi2, mlp.affine_layers, head)
    out2.backward(dout)
    close(out.detach(), out2.detach().double(), 3e-6, "output vs layer-per-launch")
    close(got_ctx, gi2.ctx2d.grad.double(), 3e-6, "d ctx vs layer-per-launch")
    # inference (no saves)
    with torch.no_grad():
        out3 = CH.chain_mlp_head(G.GatheredInput(ctx2d, idx, x), mlp.affine_layers, head)
    assert torch.equal(out3, out.detach())
