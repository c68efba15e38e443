"""Bracket time of the fused policy kernel while the rollout engine is idle vs stepping both groups (how much the
resident K1 blocks on every CU slow the other kernels of a tick down). Usage: python tools/contention_probe.py"""
import os, sys, time, threading, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egopose_amd.nets import MLP, PolicyGaussian
from egopose_amd import policy_step
from egopose_amd.hip import EgpContext
from egopose_amd.physics import SurrogatePhysics, RolloutEngine, default_threads
from egopose_amd.presets import subject_03_params
from egopose_amd.skeleton import load_skeleton
torch.manual_seed(0)
pol = PolicyGaussian(MLP(243, (300, 200), "relu"), 52, log_std=-2.3).cuda()
fp = policy_step.FusedGaussianPolicy(pol, torch.device("cuda"))
n = 512
v_out = torch.randn(n, 200, 128, device="cuda"); t_idx = torch.randint(0, 200, (n,), device="cuda")
state = torch.randn(n, 115, dtype=torch.float64, device="cuda"); noise = torch.randn(n, 52, device="cuda")
act = torch.empty(n, 52, dtype=torch.float64, device="cuda")
sk = load_skeleton(); p = subject_03_params()
ctx = EgpContext(sk, p["jkp"], p["jkd"], p["a_ref"], p["a_scale"], p["torque_lim"], p["b_diffw"], p["reward_weights"])
N = 1024
ph = SurrogatePhysics(sk, N)
eng = RolloutEngine(ctx, ph, N, n_threads=default_threads(), n_groups=2)
q0 = np.tile(np.r_[0, 0, 1.0, 1, 0, 0, 0, np.zeros(52)], (N, 1))
eng.reset(np.arange(N), q0, np.zeros((N, 58)))
action = torch.zeros(N, 52, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
stop = False
side = torch.cuda.Stream()
def stepper():
    with torch.cuda.stream(side):
        while not stop:
            for g in range(2): eng.step_async(g, action)
            for g in range(2): eng.wait(g)
def measure(tag):
    ts = []
    for i in range(80):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fp(v_out, t_idx, state, act, noise=noise); b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
        time.sleep(0.0005)
    ts = sorted(ts[10:])
    print(tag, "policy kernel bracket median", round(ts[len(ts) // 2], 1), "p90", round(ts[int(len(ts) * 0.9)], 1))
measure("engine idle   ")
th = threading.Thread(target=stepper); th.start()
time.sleep(0.2)
measure("engine running (substeps per K1 launch: %d)" % eng.substeps_per_launch)
stop = True; th.join()
torch.cuda.synchronize()
eng.close(); ph.close(); ctx.close()
