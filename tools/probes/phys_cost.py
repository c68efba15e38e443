import time, ctypes as C, numpy as np
from egopose_amd.physics import SurrogatePhysics
from egopose_amd.skeleton import load_skeleton
from egopose_amd import _lib as L
sk = load_skeleton(); P = SurrogatePhysics(sk, 64); lib = P.lib
rng = np.random.RandomState(0)
for e in range(64):
    q = np.zeros(sk.nq); q[2] = 1.0; q[3] = 1.0; q[7:] = rng.normal(size=sk.nq - 7) * 0.3
    P.reset(e, q, rng.normal(size=sk.nv) * 0.1)
ctrl = np.ascontiguousarray(rng.normal(size=sk.nu)); qp = np.empty(sk.nq); qv = np.empty(sk.nv); b = np.empty(sk.nv); xp = np.empty(3 * len(sk.body_names))
h = P.handle; N = 200000
def t(fn):
    t0 = time.perf_counter()
    for i in range(N): fn(i & 63)
    return (time.perf_counter() - t0) / N * 1e6
step = lambda e: lib.egp_physics_step_host(h, e, ctrl.ctypes.data)
dr0 = lambda e: lib.egp_physics_drain_host(h, e, qp.ctypes.data, qv.ctypes.data, None, b.ctypes.data, None)
dr1 = lambda e: lib.egp_physics_drain_host(h, e, qp.ctypes.data, qv.ctypes.data, None, b.ctypes.data, xp.ctypes.data)
nul = lambda e: lib.egp_physics_n_env(h)
cq, cv, cb, cx, cc = qp.ctypes.data, qv.ctypes.data, b.ctypes.data, xp.ctypes.data, ctrl.ctypes.data
step = lambda e: lib.egp_physics_step_host(h, e, cc)
dr0 = lambda e: lib.egp_physics_drain_host(h, e, cq, cv, None, cb, None)
dr1 = lambda e: lib.egp_physics_drain_host(h, e, cq, cv, None, cb, cx)
for name, fn in (("null call", nul), ("step", step), ("drain no xpos", dr0), ("drain + xpos (FK)", dr1)):
    print("%-20s %.3f us" % (name, min(t(fn) for _ in range(3))))
