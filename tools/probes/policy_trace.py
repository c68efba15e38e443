"""Phase stamps of one workgroup of the policy step (library built with EGP_BUILD_DEFS=-DEGP_POLICY_TRACE=<block>):
   EGP_BUILD_DEFS=-DEGP_POLICY_TRACE=3 python -m egopose_amd.build --force && python tools/probes/policy_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from egopose_amd.nets import MLP, PolicyGaussian
from egopose_amd import policy_step, _lib
torch.manual_seed(0)
pol = PolicyGaussian(MLP(243, (300, 200), "relu"), 52, log_std=-2.3).cuda()
fp = policy_step.FusedGaussianPolicy(pol, torch.device("cuda"))
n = 512
v_out = torch.randn(n, 200, 128, device="cuda"); t_idx = torch.randint(0, 200, (n,), device="cuda")
state = torch.randn(n, 115, dtype=torch.float64, device="cuda"); noise = torch.randn(n, 52, device="cuda")
act = torch.empty(n, 52, dtype=torch.float64, device="cuda")
lib = _lib.load()
lib.egp_policy_trace_read.argtypes = [ctypes.c_void_p]
names = ["start", "preload0 issued (+filter merge)", "inputs staged", "L0 k loop", "L0 partials written", "L0 barrier", "L0 reduced",
         "L1 k loop", "L1 partials", "L1 barrier", "L1 reduced", "L2 k loop", "L2 partials", "L2 barrier", "L2 reduced"]
for rep in range(6):
    fp(v_out, t_idx, state, act, noise=noise)
    torch.cuda.synchronize()
    out = np.zeros(64, np.int64)
    assert lib.egp_policy_trace_read(out.ctypes.data) == 0
    t = (out[:15] - out[0]) / 100.0
    if rep >= 4:
        print("tile", os.environ.get("EGP_POLICY_TILE", "default"), " ".join("%s %.2f" % (nm.split()[0] + nm.split()[1][:4] if len(nm.split()) > 1 else nm, x) for nm, x in zip(names, t)))
        print("   deltas us:", np.round(np.diff(t), 2).tolist())
