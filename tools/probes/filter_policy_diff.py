"""Which columns of the filtered observation differ between the two-launch filter and the apply pass inside the policy step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from egopose_amd.nets import MLP, PolicyGaussian
from egopose_amd import policy_step
from egopose_amd.hip import EgpContext
from egopose_amd.skeleton import load_skeleton
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
c = np.load(os.path.join(REPO, "tests", "golden", "config_subject_03.npz"))
ctx = EgpContext(load_skeleton(), c["jkp"], c["jkd"], c["a_ref"], c["a_scale"], c["torque_lim"], c["b_diffw"],
                 reward_weights=dict(zip([str(k) for k in c["reward_keys"]], [float(v) for v in c["reward_vals"]])), episode_len=int(c["env_episode_len"]))
torch.manual_seed(11)
n, H, S, T, nu = 512, 128, 115, 7, 52
rng = np.random.RandomState(n)
pol = PolicyGaussian(MLP(H + S, (300, 200), "relu"), nu, log_std=-2.3).cuda()
fp = policy_step.FusedGaussianPolicy(pol, torch.device("cuda"))
qpos = rng.normal(size=(n, 59)) * 0.4
qpos[:, 3:7] = rng.normal(size=(n, 4)); qpos[:, 3:7] /= np.linalg.norm(qpos[:, 3:7], axis=1, keepdims=True)
qp, qv = torch.as_tensor(qpos, device="cuda"), torch.as_tensor(rng.normal(size=(n, 58)), device="cuda")
st_a = torch.zeros(1 + 2 * S, dtype=torch.float64, device="cuda"); st_b, st_d = torch.empty_like(st_a), torch.empty_like(st_a)
v_out = torch.randn(n, T, H, device="cuda"); t_idx = torch.randint(0, T, (n,), device="cuda"); noise = torch.randn(n, nu, device="cuda")
y_ref, y2_ref = torch.empty(n, S, dtype=torch.float64, device="cuda"), torch.empty(n, S, dtype=torch.float64, device="cuda")
ctx.obs_zfilter(qp, qv, st_a, st_b, 5.0, y_ref, y2_ref)
ws = torch.empty(int(ctx.lib.egp_zfilter_workspace_bytes(n, S)) // 8, dtype=torch.float64, device="cuda")
y1, y2 = torch.zeros_like(y_ref), torch.zeros_like(y_ref)
a_f = torch.empty(n, nu, dtype=torch.float64, device="cuda")
ctx.obs_zfilter_stats(qp, qv, ws)
fp.with_filter(ctx, v_out, t_idx, qp, qv, st_a, st_d, 5.0, y1, y2, ws, a_f, noise=noise)
torch.cuda.synchronize()
d = (y1 - y_ref).abs().cpu().numpy()
print("stats equal:", torch.equal(st_b, st_d), " y equal:", bool((d == 0).all()))
cols = np.where(d.max(0) > 0)[0]
print("columns that differ:", cols.tolist())
for cc in cols[:12]:
    r = int(d[:, cc].argmax())
    print("  col %d: rows differing %d, max |diff| %.3e (row %d: %.17g vs %.17g)" % (cc, int((d[:, cc] > 0).sum()), d[:, cc].max(), r, y1[r, cc].item(), y_ref[r, cc].item()))
