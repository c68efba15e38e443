#!/usr/bin/env python3
"""Throughput of one state_reg optimisation step (BASELINE config 4: VideoRegNet = ResNet-18 -> bi-LSTM -> MLP on optical-flow
clips of 224x224 frames), fp32 and bf16 autocast. Usage: python tools/statereg_bench.py [frames_per_clip]"""
import os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egopose_amd.nets import VideoRegNet

T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
# (torch.backends.cudnn.benchmark = True -- MIOpen's exhaustive search -- was tried: six minutes of tuning, same step time)
dev = torch.device("cuda")
torch.manual_seed(0)
for name, ac, cl in (("fp32 channels_last (default)", None, True), ("fp32 NCHW", None, False),
                     ("bf16 autocast channels_last", torch.bfloat16, True), ("bf16 autocast NCHW", torch.bfloat16, False)):
    net = VideoRegNet(115, 128, 128, no_cnn=False).to(dev)
    if cl:
        net.channels_last()       # what StateRegTrainer does on the GPU
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    x = torch.randn(T, 1, 3, 224, 224, device=dev)
    gt = torch.randn(T, 115, device=dev)
    def step():
        if ac is not None:
            with torch.autocast("cuda", dtype=ac):
                pred = net(x).float()
        else:
            pred = net(x)
        loss = (gt - pred).pow(2).sum(1).mean()
        opt.zero_grad(); loss.backward(); opt.step()
        return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.time()
    n = 8
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print("%-28s clip of %d frames: %.1f ms per step, %.0f frames/s" % (name, T, dt * 1e3, T / dt))
