#!/usr/bin/env python3
"""Kernel microbenchmarks K1-K6 at several env counts (HIP events on the launch stream).
Prints one JSON line per (kernel, n): time per launch, algorithmic bytes (float64),
achieved GB/s and fraction of the 8 TB/s HBM peak."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egopose_amd.hip import EgpContext
from egopose_amd.skeleton import load_skeleton
from egopose_amd.presets import subject_03_params

HBM_PEAK = 8.0e12


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    sk = load_skeleton()
    p = subject_03_params()
    ctx = EgpContext(sk, p["jkp"], p["jkd"], p["a_ref"], p["a_scale"], p["torque_lim"], p["b_diffw"], p["reward_weights"])
    sizes = [int(s) for s in (sys.argv[1:] or ["1024", "8192", "65536"])]
    dt = torch.float64
    W = 8
    rng = np.random.RandomState(0)
    M0 = sk.zero_pose_inertia()
    qM0 = sk.sparse_from_full(M0)
    # a small expert table for the reward gathers
    F = 4096
    take = dict(qpos=rng.normal(size=(F, 59)), qvel=rng.normal(size=(F, 58)), rlinv_local=rng.normal(size=(F, 3)),
                rangv=rng.normal(size=(F, 3)), rq_rmh=rng.normal(size=(F, 4)), ee_pos=rng.normal(size=(F, 15)),
                bquat=rng.normal(size=(F, 84)), bangvel=rng.normal(size=(F, 63)), head_height_lb=1.0)
    ctx.upload_experts([take])
    for n in sizes:
        g = lambda *s: torch.randn(*s, dtype=dt, device="cuda")
        qpos, qvel, act, C = g(n, 59) * 0.3, g(n, 58), g(n, 52) * 0.3, g(n, 58)
        qpos[:, 3:7] = torch.nn.functional.normalize(g(n, 4), dim=1)
        prev = qpos + g(n, 59) * 0.01
        qM = torch.as_tensor(qM0, device="cuda").repeat(n, 1).contiguous()
        ee = g(n, 15)
        t = torch.randint(1, 100, (n,), dtype=torch.int32, device="cuda")
        frame = torch.randint(0, F, (n,), dtype=torch.int32, device="cuda")
        end = torch.zeros(n, dtype=torch.int32, device="cuda")
        obs = g(n, 115)
        qM_dyn = torch.empty(n, sk.nM, dtype=dt, device="cuda")
        st0 = torch.zeros(231, dtype=dt, device="cuda")
        st1 = torch.empty_like(st0)
        rew, msk, val = torch.rand(n * 200 // 8, dtype=dt, device="cuda"), torch.ones(n * 200 // 8, dtype=dt, device="cuda"), g(n * 200 // 8)
        cases = [
            ("K1_pd_torque", lambda: ctx.pd_torque(qpos, qvel, act, qM, C), (910 + 58 + 52 + 58 + 52 + 52) * W, 1),
            ("K2_reward", lambda: ctx.reward(qpos, prev, ee, t, frame, end, 0.0), (59 + 59 + 15 + 166 + 6) * W, 1),
            ("K3_obs", lambda: ctx.obs(qpos, qvel), (59 + 58 + 115) * W, 1),
            ("K4_body_quat", lambda: ctx.body_quat(qpos), (59 + 84) * W, 1),
            ("K6_zfilter", lambda: ctx.zfilter(obs, st0, st1, update=True), (115 + 115) * W, 1),
            ("K5_gae", lambda: ctx.gae(rew, msk, val, 0.95, 0.95), 5 * W, rew.shape[0] / n),
            # K8: reads qpos 59 + qvel 58, writes qM 910 + bias 58 + xpos 63
            ("K8_dynamics", lambda: ctx.dynamics(qpos, qvel, want_xpos=True, qM_out=qM_dyn), (59 + 58 + 910 + 58 + 63) * W, 1),
        ]
        for name, fn, bytes_per_unit, units_per_env in cases:
            if name == "K1_pd_torque":
                for variant in (0, 2, 1):
                    ctx.set_pd_variant(variant)
                    s = timeit(fn, iters=20 if variant == 1 else 50)
                    ab = bytes_per_unit * n
                    print(json.dumps(dict(kernel=name + {0: "_tree58", 2: "_reg58", 1: "_lds"}[variant], n=n, us=s * 1e6, alg_bytes=ab,
                                          GBps=ab / s / 1e9, frac_hbm=ab / s / HBM_PEAK)))
                ctx.set_pd_variant(0)
                continue
            s = timeit(fn)
            ab = bytes_per_unit * n * units_per_env
            print(json.dumps(dict(kernel=name, n=n, us=s * 1e6, alg_bytes=ab, GBps=ab / s / 1e9, frac_hbm=ab / s / HBM_PEAK)))


if __name__ == "__main__":
    main()
