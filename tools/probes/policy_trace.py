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
FILTER = os.environ.get("EGP_TRACE_FILTER") == "1"        # the rollout's form: the filter's apply pass in the prologue
if FILTER:
    from egopose_amd.hip import EgpContext
    from egopose_amd.skeleton import load_skeleton
    REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    c = np.load(os.path.join(REPO, "tests", "golden", "config_subject_03.npz"))
    ctx = EgpContext(load_skeleton(), c["jkp"], c["jkd"], c["a_ref"], c["a_scale"], c["torque_lim"], c["b_diffw"],
                     reward_weights=dict(zip([str(k) for k in c["reward_keys"]], [float(v) for v in c["reward_vals"]])),
                     episode_len=int(c["env_episode_len"]))
    rng = np.random.RandomState(0)
    qpos = rng.normal(size=(n, 59)) * 0.4
    qpos[:, 3:7] = rng.normal(size=(n, 4)); qpos[:, 3:7] /= np.linalg.norm(qpos[:, 3:7], axis=1, keepdims=True)
    qp, qv = torch.as_tensor(qpos, device="cuda"), torch.as_tensor(rng.normal(size=(n, 58)), device="cuda")
    st_a = torch.zeros(231, dtype=torch.float64, device="cuda"); st_b = torch.empty_like(st_a)
    y1, y2 = torch.empty(n, 115, dtype=torch.float64, device="cuda"), torch.empty(n, 115, dtype=torch.float64, device="cuda")
    ws = torch.empty(int(ctx.lib.egp_zfilter_workspace_bytes(n, 115)) // 8, dtype=torch.float64, device="cuda")
    ctx.obs_zfilter_stats(qp, qv, ws)
for rep in range(6):
    if FILTER:
        fp.with_filter(ctx, v_out, t_idx, qp, qv, st_a, st_b, 5.0, y1, y2, ws, act, noise=noise)
    else:
        fp(v_out, t_idx, state, act, noise=noise)
    torch.cuda.synchronize()
    out = np.zeros(64, np.int64)
    assert lib.egp_policy_trace_read(out.ctypes.data) == 0
    t = (out[:15] - out[0]) / 100.0
    if rep >= 4:
        print("filter" if FILTER else "plain", "tile", os.environ.get("EGP_POLICY_TILE", "default"), " ".join("%s %.2f" % (nm.split()[0] + nm.split()[1][:4] if len(nm.split()) > 1 else nm, x) for nm, x in zip(names, t)))
        print("   deltas us:", np.round(np.diff(t), 2).tolist())
        if FILTER:
            print("   state wave (thread 128), us from start: merge begins / merged / 1/std / rows computed / rows stored / LDS written:",
                  np.round((out[32:38] - out[0]) / 100.0, 2).tolist())
