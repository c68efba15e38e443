#!/usr/bin/env python3
"""Wall time of one env-step of the rollout engine (both groups stepping concurrently, no policy / reward work) as a
function of how many envs are still active -- the tail of a rollout is bound by this. Usage: python tools/engine_step_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egopose_amd.hip import EgpContext
from egopose_amd.physics import SurrogatePhysics, RolloutEngine, default_threads
from egopose_amd.presets import subject_03_params
from egopose_amd.skeleton import load_skeleton

N, G = 1024, 2
sk = load_skeleton(); p = subject_03_params()
ctx = EgpContext(sk, p["jkp"], p["jkd"], p["a_ref"], p["a_scale"], p["torque_lim"], p["b_diffw"], p["reward_weights"])
ph = SurrogatePhysics(sk, N)
eng = RolloutEngine(ctx, ph, N, n_threads=default_threads(), n_groups=G)
q0 = np.tile(np.r_[0, 0, 1.0, 1, 0, 0, 0, np.zeros(52)], (N, 1))
eng.reset(np.arange(N), q0, np.zeros((N, 58)))
act = torch.zeros(N, 52, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
rng = np.random.RandomState(0)
print("threads", eng.n_threads, "substeps per K1 launch", eng.substeps_per_launch)
for n_active in (1024, 512, 256, 64, 16, 2):
    mask = np.zeros(N, np.int32)
    mask[rng.choice(N, n_active, replace=False)] = 1
    for it in range(5):
        for g in range(G): eng.step_async(g, act, mask)
        for g in range(G): eng.wait(g)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 40
    for it in range(reps):
        for g in range(G): eng.step_async(g, act, mask)
        for g in range(G): eng.wait(g)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("active %5d: %.0f us per env-step (%.1f us per substep)" % (n_active, dt * 1e6, dt * 1e6 / 15))
eng.close(); ph.close(); ctx.close()
