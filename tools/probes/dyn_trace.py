"""Phase stamps of one wave of K8 (library built with EGP_BUILD_DEFS=-DEGP_DYN_TRACE): python tools/probes/dyn_trace.py [n_env]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from egopose_amd import _lib
from egopose_amd.hip import EgpContext
from egopose_amd.presets import subject_03_params
from egopose_amd.skeleton import load_skeleton
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sk = load_skeleton(); p = subject_03_params()
ctx = EgpContext(sk, p["jkp"], p["jkd"], p["a_ref"], p["a_scale"], p["torque_lim"], p["b_diffw"], p["reward_weights"])
rng = np.random.RandomState(0)
q = rng.normal(size=(n, 59)) * 0.3; q[:, 3:7] = rng.normal(size=(n, 4)); q[:, 3:7] /= np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
qd = torch.as_tensor(q, device="cuda"); vd = torch.as_tensor(rng.normal(size=(n, 58)), device="cuda")
lib = _lib.load(); lib.egp_dyn_trace_read.argtypes = [ctypes.c_void_p]
for rep in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); out = ctx.dynamics(qd, vd); b.record(); torch.cuda.synchronize()
    t = np.zeros(16, np.int64); lib.egp_dyn_trace_read(t.ctypes.data)
    d = np.diff(t[:6]) / 100.0
    print("n %d  kernel bracket %.1f us   phases us: A %.2f  B %.2f  C %.2f  D %.2f  E+F %.2f (E %.2f)   total %.2f" % (n, a.elapsed_time(b) * 1e3, *d, (t[6] - t[4]) / 100.0, d.sum()))
