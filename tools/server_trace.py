#!/usr/bin/env python3
"""Timeline of one env-step in resident-K1 mode (EGP_SERVER_TRACE=1): block 0's device stamps and the host stamps of
the thread that owns slice 0, per substep. Usage: python tools/server_trace.py [n_env] [n_groups]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ["EGP_SERVER_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egopose_amd import _lib as L
from egopose_amd.hip import EgpContext
from egopose_amd.physics import SurrogatePhysics, RolloutEngine, default_threads
from egopose_amd.presets import subject_03_params
from egopose_amd.skeleton import load_skeleton


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    groups = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    n_active = int(sys.argv[3]) if len(sys.argv) > 3 else n          # env 0 (block 0 / slice 0, the traced ones) stays active
    sk = load_skeleton()
    p = subject_03_params()
    ctx = EgpContext(sk, p["jkp"], p["jkd"], p["a_ref"], p["a_scale"], p["torque_lim"], p["b_diffw"], p["reward_weights"])
    ph = SurrogatePhysics(sk, n)
    dyn = os.environ.get("EGP_DEVICE_DYNAMICS", "0") == "1"          # K8 inside the resident kernel (stamp 5: K8 + factors done)
    eng = RolloutEngine(ctx, ph, n, n_threads=default_threads(), n_groups=groups, device_dynamics=dyn)
    rng = np.random.RandomState(0)
    q0 = np.tile(sk.default_qpos() if hasattr(sk, "default_qpos") else np.r_[0, 0, 1.0, 1, 0, 0, 0, np.zeros(52)], (n, 1))
    eng.reset(np.arange(n), q0, np.zeros((n, 58)))
    act = torch.as_tensor(rng.normal(size=(n, 52)) * 0.1, device="cuda")
    torch.cuda.synchronize()
    mask = None
    if n_active < n:
        mask = np.zeros(n, np.int32)
        mask[0] = 1
        mask[np.random.RandomState(1).choice(np.arange(1, n), max(0, n_active - 1), replace=False)] = 1
    import time
    t_steps = []
    for it in range(20):
        t0 = time.perf_counter()
        for g in range(groups):
            eng.step_async(g, act, mask)
        for g in range(groups):
            eng.wait(g)
        t_steps.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    print("env-step wall time (us), last 5:", [round(t) for t in t_steps[-5:]])
    fs = 15
    dev = np.zeros(fs * 8, np.int64)
    host = np.zeros(fs * 4)
    L.check(eng.lib.egp_engine_server_trace(eng.handle, 0, dev.ctypes.data, host.ctypes.data), "trace")
    dev = dev.reshape(fs, 8).astype(float) / 100.0        # us
    host = host.reshape(fs, 4)
    t0 = dev[0, 0]
    print("device (block 0), us since kernel start:  poll | go seen | loaded | solved | stored")
    for s in range(fs):
        r = dev[s] - t0
        print("  sub %2d  %8.2f %8.2f %8.2f %8.2f %8.2f   | wait %.2f load %.2f solve %.2f store %.2f%s" % (
            s, r[0], r[1], r[2], r[3], r[4], r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3],
            ("  then (last env of the wave stored / K8 + factors done) %.2f" % (r[5] - r[4])) if dev[s, 5] > 0 else ""))
    print("host (owner of slice 0), us since the step was posted:  wait start | first row in | go written")
    for s in range(fs):
        print("  sub %2d  %8.2f %8.2f %8.2f   | waited %.2f physics %.2f" % (s, host[s, 0], host[s, 1], host[s, 2], host[s, 1] - host[s, 0], host[s, 2] - host[s, 1]))
    print("leader, us since the step was posted: launch issued %.1f | own substeps done %.1f | all threads done %.1f | kernel done %.1f" % tuple(host[:4, 3]))
    print("per-thread finish (us since posted; each thread's own clock start):", [round(x) for x in host[4:, 3] if x > 0])
    print("envs per wave", eng.envs_per_wave, "resident capacity", eng.resident_capacity)
    print("threads", eng.n_threads if hasattr(eng, "n_threads") else "?", "timing", eng.timing())
    eng.close(); ph.close(); ctx.close()


if __name__ == "__main__":
    main()
