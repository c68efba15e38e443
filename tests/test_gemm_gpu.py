"""csrc/egp_gemm.hip through the C-ABI (`egp_gemm_f32`) against float64 products: the four operand layouts, ragged sizes
(tiles cut in m, n and k; rows that are only 4-byte aligned), the fused epilogues, split-K with the bias-gradient column,
and the autograd node that the update's MLPs run on (against torch's own float32 autograd of the reference's modules)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["ws", "classic"], autouse=True)
def three_piece_kernel(request, monkeypatch):
    """terms = 6 runs on the warp-specialised persistent kernel (k_gemm_ws) where its preconditions hold (k ranges of at
    least one k-tile, 16-byte-friendly outputs) and on k_gemm_bf16x otherwise; EGP_GEMM_WS=0 sends everything to the latter.
    Every test of this file runs both ways."""
    monkeypatch.setenv("EGP_GEMM_WS", "1" if request.param == "ws" else "0")
    return request.param


def _rel(got, ref):
    return float((got.double() - ref).norm() / ref.norm().clamp_min(1e-300))


def _operands(M, N, K, a_kc, b_kc, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    A64 = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
    B64 = torch.randn(K, N, dtype=torch.float64, device="cuda", generator=g)
    A = (A64 if a_kc else A64.t()).float().contiguous()          # (M, K) or (K, M)
    B = (B64.t() if b_kc else B64).float().contiguous()          # (N, K) or (K, N)
    A64 = (A if a_kc else A.t()).double()
    B64 = (B.t() if b_kc else B).double()
    return A, B, A64 @ B64


@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("M,N,K", [(1000, 300, 243), (257, 200, 300), (129, 52, 200), (64, 1, 77), (5, 243, 1), (300, 243, 4097),
                                   (128, 128, 32), (1, 1, 1), (3000, 244, 300), (700, 128, 33), (20000, 1024, 128)])
def test_products_match_float64(a_kc, b_kc, M, N, K):
    from egopose_amd.gemm import gemm
    A, B, ref = _operands(M, N, K, a_kc, b_kc, seed=M + 7 * N + 13 * K)
    c6 = gemm(A, B, a_kc, b_kc, terms=6)
    c3 = gemm(A, B, a_kc, b_kc, terms=3)
    c1 = gemm(A, B, a_kc, b_kc, terms=1)
    assert c3.shape == (M, N) and c6.shape == (M, N)
    assert _rel(c3, ref) < 2e-5, "split-operand product must be float32-class"
    assert _rel(c1, ref) < 6e-3
    # element-wise: 2^-15 of the sum of the magnitudes of the K products
    A64 = (A if a_kc else A.t()).double().abs()
    B64 = (B.t() if b_kc else B).double().abs()
    assert ((c3.double() - ref).abs() <= 3.1e-5 * (A64 @ B64) + 1e-30).all()
    lib = A.double() if a_kc else A.t().double()
    f32 = (lib.float() @ (B.t() if b_kc else B)).double()          # the library's float32 product, for scale
    assert _rel(c3, ref) < max(40 * _rel(f32, ref), 1.6e-5)
    # three-piece operands: float32-class (the library's float32 product is the yardstick)
    assert _rel(c6, ref) < max(3 * _rel(f32, ref), 2e-7), (_rel(c6, ref), _rel(f32, ref))
    assert ((c6.double() - ref).abs() <= 8e-7 * (A64 @ B64) + 1e-30).all()


def test_rows_with_4_byte_alignment_and_strided_views():
    """Operands that are column slices of wider tensors (leading dimension != width, rows not 16-byte aligned)."""
    from egopose_amd.gemm import gemm
    g = torch.Generator(device="cuda").manual_seed(3)
    big = torch.randn(700, 251, device="cuda", generator=g)
    W = torch.randn(97, 131, device="cuda", generator=g)
    x = big[:, 3:134]                                       # (700, 131), ld 251, offset 3 floats
    ref = x.double() @ W.double().t()
    assert _rel(gemm(x, W), ref) < 2e-5
    out = torch.zeros(700, 120, device="cuda")
    gemm(x, W, out=out[:, 11:108])                          # strided output
    assert _rel(out[:, 11:108], ref) < 2e-5 and float(out[:, :11].abs().sum() + out[:, 108:].abs().sum()) == 0.0
    dy = big[:, 100:197]                                    # (700, 97)
    dW, db = gemm(dy, x, False, False, splits=5, want_bias_grad=True)
    assert _rel(dW, dy.double().t() @ x.double()) < 2e-5 and _rel(db, dy.double().sum(0)) < 2e-5


def test_epilogues():
    from egopose_amd.gemm import gemm, linear_dgrad, linear_fwd
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(900, 243, device="cuda", generator=g)
    W = torch.randn(300, 243, device="cuda", generator=g) * 0.1
    b = torch.randn(300, device="cuda", generator=g)
    z = x.double() @ W.double().t() + b.double()
    y = linear_fwd(x, W, b, relu=True)
    assert _rel(y, z.clamp_min(0)) < 2e-5
    sure = z.abs() > 1e-4                                     # the sign of near-zero pre-activations may differ in float32
    assert ((y > 0) == (z > 0))[sure].all()
    assert _rel(linear_fwd(x, W, b), z) < 2e-5 and _rel(linear_fwd(x, W), z - b.double()) < 2e-5
    dy = torch.randn(900, 300, device="cuda", generator=g)
    h = torch.randn(900, 243, device="cuda", generator=g)
    ref = (dy.double() @ W.double()) * (h > 0)
    assert _rel(linear_dgrad(dy, W, mask=h), ref) < 2e-5
    assert _rel(linear_dgrad(dy, W, n_cols=128), dy.double() @ W.double()[:, :128]) < 2e-5


@pytest.mark.parametrize("splits", [1, 2, 7, 64])
def test_split_k_weight_gradient_with_bias_column(splits):
    from egopose_amd.gemm import gemm, linear_wgrad, pick_splits
    g = torch.Generator(device="cuda").manual_seed(11)
    n = 9001
    dy = torch.randn(n, 300, device="cuda", generator=g)
    x = torch.randn(n, 243, device="cuda", generator=g)
    dW_ref, db_ref = dy.double().t() @ x.double(), dy.double().sum(0)
    dW, db = gemm(dy, x, False, False, splits=splits, want_bias_grad=True)
    assert dW.shape == (300, 243) and db.shape == (300,)
    assert _rel(dW, dW_ref) < 2e-5 and _rel(db, db_ref) < 2e-5
    dW2, db2 = gemm(dy, x, False, False, splits=splits, want_bias_grad=True)
    assert torch.equal(dW, dW2) and torch.equal(db, db2), "fixed-order reduction: bit-identical from run to run"
    acc, accb = dW.clone(), db.clone()
    gemm(dy, x, False, False, splits=splits, want_bias_grad=True, out=acc, bias_grad_out=accb, accumulate=True)
    assert _rel(acc, 2 * dW_ref) < 2e-5 and _rel(accb, 2 * db_ref) < 2e-5
    if splits == 1:
        a, b2 = linear_wgrad(dy, x)
        assert pick_splits(300, 244, n) > 1 and _rel(a, dW_ref) < 2e-5 and _rel(b2, db_ref) < 2e-5
        assert _rel(gemm(dy, x, False, False, splits=3), dW_ref) < 2e-5          # split-K without the bias column


def test_bad_arguments_are_refused():
    from egopose_amd.gemm import gemm
    a, b = torch.zeros(8, 4, device="cuda"), torch.zeros(5, 4, device="cuda")
    with pytest.raises(ValueError):
        gemm(a, torch.zeros(5, 3, device="cuda"))
    with pytest.raises(ValueError):
        gemm(a.double(), b.double())
    with pytest.raises(ValueError):
        gemm(a, b, terms=2)
    with pytest.raises(ValueError):
        gemm(a, b, terms=4)
    with pytest.raises(ValueError):
        gemm(a, b, splits=2, bias=torch.zeros(5, device="cuda"))
    with pytest.raises(ValueError):
        gemm(a.t(), b)                      # second dimension not unit-stride
    assert gemm(torch.zeros(0, 4, device="cuda"), b).shape == (0, 5)


@pytest.mark.parametrize("head_dim", [52, 1])
def test_mlp_head_node_matches_torch_autograd(head_dim):
    """The fused node (forward + every gradient) against torch autograd over the reference's modules in float64."""
    from egopose_amd.gemm import mlp_head
    from egopose_amd.nets import MLP
    torch.manual_seed(0)
    n = 5000
    net = MLP(243, [300, 200], "relu").cuda()
    head = torch.nn.Linear(200, head_dim).cuda()
    x = torch.randn(n, 243, device="cuda", requires_grad=True)
    tgt = torch.randn(n, head_dim, device="cuda")
    out = mlp_head(x, net.affine_layers, head, n_grad_cols=128)
    loss = (out - tgt).pow(2).mean()
    loss.backward()
    got = [x.grad.clone()] + [p.grad.clone() for p in list(net.parameters()) + list(head.parameters())]
    net64, head64 = MLP(243, [300, 200], "relu").cuda().double(), torch.nn.Linear(200, head_dim).cuda().double()
    net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    head64.load_state_dict({k: v.double() for k, v in head.state_dict().items()})
    x64 = x.detach().double().requires_grad_(True)
    out64 = head64(net64(x64))
    (out64 - tgt.double()).pow(2).mean().backward()
    ref = [x64.grad] + [p.grad for p in list(net64.parameters()) + list(head64.parameters())]
    assert _rel(out.detach(), out64.detach()) < 2e-5
    assert float(got[0][:, 128:].abs().max()) == 0.0
    for a, b in zip(got[1:], ref[1:]):
        assert a.shape == b.shape
    # the torch float32 path of the same modules, for scale. (Gradients that pass through a ReLU derivative differ from
    # float64 in ANY float32 evaluation wherever a pre-activation is within round-off of zero and the derivative flips:
    # ~1e-3 in norm for the library path too; the number of such units grows with the product's round-off.)
    for p in list(net.parameters()) + list(head.parameters()):
        p.grad = None
    x2 = x.detach().clone().requires_grad_(True)
    (head(net(x2)) - tgt).pow(2).mean().backward()
    lib = [x2.grad] + [p.grad for p in list(net.parameters()) + list(head.parameters())]
    assert _rel(got[0][:, :128], ref[0][:, :128]) < max(10 * _rel(lib[0][:, :128], ref[0][:, :128]), 5e-5)
    assert _rel(got[0][:, :128], lib[0][:, :128].double()) < 5e-3
    for a, l, b in zip(got[1:], lib[1:], ref[1:]):
        assert _rel(a, b) < max(10 * _rel(l, b), 5e-5), (a.shape, _rel(a, b), _rel(l, b))
        assert _rel(a, l.double()) < 5e-3
    # without activation-derivative flips in the way (the head layer sees none): float32-class
    assert _rel(got[-2], ref[-2]) < 5e-5 and _rel(got[-1], ref[-1]) < 5e-5


def test_gather_concat_node_matches_torch():
    """`egp_gather_concat_f32` / `egp_scatter_rows_f32` (the update's policy input, models/video_state_net.py:65-69)
    against index_select + cat under torch autograd, bit for bit (pure data movement)."""
    from egopose_amd.gemm import GatherConcat, gather_concat_available
    g = torch.Generator(device="cuda").manual_seed(2)
    R, H, S, n = 977, 128, 115, 611
    ctx = torch.randn(R, H, device="cuda", generator=g, requires_grad=True)
    x = torch.randn(n, S, device="cuda", generator=g)
    idx = torch.randperm(R, device="cuda", generator=g)[:n].contiguous()
    assert gather_concat_available(ctx, idx, x)
    out = GatherConcat.apply(ctx, idx, x)
    w = torch.randn(n, H + S, device="cuda", generator=g)
    (out * w).sum().backward()
    got = ctx.grad.clone()
    ctx.grad = None
    ref = torch.cat((ctx.index_select(0, idx), x), 1)
    (ref * w).sum().backward()
    assert torch.equal(out, ref) and torch.equal(got, ctx.grad)
    assert not gather_concat_available(ctx, idx, x.clone().requires_grad_(True))       # state gradients: the torch path


def test_fused_gather_scatter_operands(three_piece_kernel):
    """a_rows / a2 (first layer reads [ctx[idx] | state]), b_krows / b2 (its weight gradient) and c_rows (its data gradient
    writes the context rows) against the same products on materialised operands."""
    from egopose_amd.gemm import gemm
    g = torch.Generator(device="cuda").manual_seed(21)
    R, n, H, S, N1 = 5000, 4321, 128, 115, 300
    ctx2d = torch.randn(R, H, device="cuda", generator=g)
    state = torch.randn(n, S, device="cuda", generator=g)
    idx = torch.randperm(R, device="cuda", generator=g)[:n].contiguous()
    W = torch.randn(N1, H + S, device="cuda", generator=g) * 0.1
    b = torch.randn(N1, device="cuda", generator=g)
    x = torch.cat((ctx2d[idx], state), 1)
    dz = torch.randn(n, N1, device="cuda", generator=g)
    if three_piece_kernel == "classic":          # the classic kernel has no gather path: refused, not silently wrong
        with pytest.raises(ValueError):
            gemm(ctx2d, W, True, True, bias=b, relu=True, a_rows=idx, a2=state)
        return
    y = gemm(ctx2d, W, True, True, bias=b, relu=True, a_rows=idx, a2=state)
    assert torch.equal(y, gemm(x, W, True, True, bias=b, relu=True)), "same products, same order: bit-identical to the materialised input"
    dW, db = gemm(dz, ctx2d, False, False, splits=7, want_bias_grad=True, b_krows=idx, b2=state)
    dW_ref, db_ref = gemm(dz, x, False, False, splits=7, want_bias_grad=True)
    assert dW.shape == (N1, H + S) and torch.equal(dW, dW_ref) and torch.equal(db, db_ref)
    with pytest.raises(ValueError):              # fewer than one k-tile of second-source columns: a k-tile would straddle the sources
        gemm(ctx2d, W[:, :H + 20].contiguous(), True, True, a_rows=idx, a2=state[:, :20].contiguous())
    dctx = torch.zeros(R, H, device="cuda")
    gemm(dz, W[:, :H], True, False, out=dctx, c_rows=idx)
    ref = torch.zeros(R, H, device="cuda")
    ref[idx] = gemm(dz, W[:, :H], True, False)
    assert torch.equal(dctx, ref)
    assert _rel(y, (x.double() @ W.double().t() + b.double()).clamp_min(0)) < 2e-5


def test_gather_mlp_head_node_matches_the_two_node_form(three_piece_kernel):
    from egopose_amd import gemm as G
    if three_piece_kernel == "classic":
        assert not G.fused_gather_available(128, 300, 115)
        return
    assert G.fused_gather_available(128, 300, 115) and not G.fused_gather_available(100, 300, 115) and not G.fused_gather_available(128, 300, 20)
    g = torch.Generator(device="cuda").manual_seed(22)
    R, n, H, S = 3000, 2500, 128, 115
    layers = torch.nn.ModuleList([torch.nn.Linear(H + S, 300), torch.nn.Linear(300, 200)]).cuda()
    head = torch.nn.Linear(200, 52).cuda()
    idx = torch.randperm(R, device="cuda", generator=g)[:n].contiguous()
    state = torch.randn(n, S, device="cuda", generator=g)
    w = torch.randn(n, 52, device="cuda", generator=g)
    outs = []
    for fused in (True, False):
        ctx2d = torch.randn(R, H, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)).requires_grad_()
        for p in list(layers.parameters()) + list(head.parameters()):
            p.grad = None
        if fused:
            out = G.gather_mlp_head(G.GatheredInput(ctx2d, idx, state), layers, head)
        else:
            out = G.mlp_head(G.GatherConcat.apply(ctx2d, idx, state), layers, head, H)
        (out * w).sum().backward()
        outs.append([out.detach().clone(), ctx2d.grad.clone()] + [p.grad.clone() for p in list(layers.parameters()) + list(head.parameters())])
    for a, b2 in zip(*outs):
        assert torch.equal(a, b2)
