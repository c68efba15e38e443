#!/usr/bin/env python3
"""Throughput of one state_reg optimisation step (BASELINE config 4: VideoRegNet = ResNet-18 -> bi-LSTM -> MLP on optical-flow
clips of 224x224 frames): fp32, bf16 autocast and the bf16 encoder with fp32 master weights (the default on the GPU).
Usage: python tools/statereg_bench.py [frames_per_clip] [variant-name filter]"""
import os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egopose_amd.nets import VideoRegNet

T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""          # substring filter on the variant name
# (torch.backends.cudnn.benchmark = True -- MIOpen's exhaustive search -- was tried: six minutes of tuning, same step time)
dev = torch.device("cuda")
torch.manual_seed(0)
for name, ac, cl in (("fp32 channels_last (default)", None, True), ("fp32 NCHW", None, False),
                     ("bf16 autocast channels_last", torch.bfloat16, True), ("bf16 autocast NCHW", torch.bfloat16, False),
                     ("bf16 encoder (Bf16Shadow, fp32 masters) channels_last", "enc", True),
                     ("bf16 encoder (Bf16Shadow, fp32 masters) NCHW", "enc", False)):
    if ONLY and ONLY not in name:
        continue
    net = VideoRegNet(115, 128, 128, no_cnn=False).to(dev)
    if cl:
        net.channels_last()       # what StateRegTrainer does on the GPU
    if ac == "enc":
        net.bf16_encoder()          # what StateRegTrainer does on the GPU (BASELINE config 4)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    x = torch.randn(T, 1, 3, 224, 224, device=dev)
    gt = torch.randn(T, 115, device=dev)
    def step():
        if ac == "enc":
            pred = net(x)
        elif ac is not None:
            with torch.autocast("cuda", dtype=ac):
                pred = net(x).float()
        else:
            pred = net(x)
        loss = (gt - pred).pow(2).sum(1).mean()
        opt.zero_grad(); loss.backward(); net.encoder_grads_ready(); opt.step(); net.encoder_stepped()
        return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.time()
    n = 8
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print("%-28s clip of %d frames: %.1f ms per step, %.0f frames/s" % (name, T, dt * 1e3, T / dt))
