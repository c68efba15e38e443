"""BASELINE config 4 on the device: "state_reg cross_01: ResNet-18 VideoRegNet bf16 on MFMA, batch 256, 1xMI355X" at its own
shape -- one clip of 256 optical-flow frames of 224 x 224 per optimisation step (models/video_reg_net.py:10-59,
ego_pose/state_reg.py:60-95)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_config4_shape_optimisation_step_bf16_encoder():
    """One optimisation step of VideoRegNet at batch 256 x 224 x 224 with the encoder as a bf16 copy (float32 masters):
    finite loss, every master parameter moves, the bf16 copy follows the masters, the bf16 forward agrees with the float32
    forward of the same weights and batch statistics, and what is kept (eval / feature export) is float32."""
    from egopose_amd.nets import VideoRegNet
    torch.manual_seed(11)
    net = VideoRegNet(115, 128, 128, no_cnn=False).cuda().channels_last()
    assert net.frame_shape == (3, 224, 224)
    x = torch.randn(256, 1, 3, 224, 224, device="cuda")
    gt = torch.randn(256, 115, device="cuda")
    net.train()
    net32 = copy.deepcopy(net)
    net.bf16_encoder()
    with torch.no_grad():
        ref = net32(x)                                # float32 encoder, train-mode batch statistics
    pred = net(x)                                     # bf16 encoder on the matrix cores
    assert pred.shape == (256, 115) and pred.dtype == torch.float32
    rel = float((pred.detach() - ref).norm() / ref.norm())
    assert 1e-5 < rel < 3e-2, rel
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4)
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    loss = (gt - pred).pow(2).sum(dim=1).mean()
    opt.zero_grad()
    loss.backward()
    net.encoder_grads_ready()
    opt.step()
    net.encoder_stepped()
    assert np.isfinite(float(loss.detach()))
    for n, p in net.named_parameters():
        assert p.dtype == torch.float32 and torch.isfinite(p).all() and not torch.equal(p, before[n]), n
    for m, s in net._enc16.pairs:
        assert s.dtype == torch.bfloat16 and torch.equal(s, m.to(torch.bfloat16))
    # a second step lowers the loss on the same clip (the gradients that went through the bf16 copy point downhill)
    loss2 = (gt - net(x)).pow(2).sum(dim=1).mean()
    assert float(loss2) < float(loss)
    net.eval()
    with torch.no_grad():
        feats = net.get_cnn_feature(x[:32])
        want = net.cnn(net._frames(x[:32]))            # (two MIOpen float32 calls agree to round-off, not bit for bit)
        assert feats.shape == (32, 128) and float((feats - want).norm() / want.norm()) < 1e-5


def test_config4_bench_leg_reports_frames_per_second():
    from egopose_amd.bench_support import statereg_config4
    r = statereg_config4(0, frames=256, steps=2, warmup=1)
    assert r["frames_per_step"] == 256 and r["frame_shape"] == [3, 224, 224] and r["frames_per_s"] > 0 and np.isfinite(r["loss_last"])
    assert r["parameters"] > 11_176_512
