import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
from conftest import load_golden
from update_fixture import build_agent, batch_of
from egopose_amd.hip import EgpContext
from egopose_amd.skeleton import load_skeleton
g = load_golden("ppo_update_h128.npz"); c = load_golden("config_subject_03.npz")
ctx = EgpContext(load_skeleton(), c["jkp"], c["jkd"], c["a_ref"], c["a_scale"], c["torque_lim"], c["b_diffw"])
for mode in ("hip", "torch"):
    os.environ["EGP_GEMM"] = mode
    agent, mods = build_agent(g, device="cuda", dtype=torch.float32, fused_adam=True)
    feats = [np.asarray(g["cnn_feat0"], np.float64), np.asarray(g["cnn_feat1"], np.float64)]
    table = torch.as_tensor(np.concatenate(feats, 0), dtype=torch.float32, device="cuda")
    for net in (agent.cn.policy_vs_net, agent.cn.value_vs_net):
        net.attach_feature_table(table, [0, feats[0].shape[0]])
    agent._kernel_ctx = lambda: ctx
    agent.update_params(batch_of(g))
    worst = 0; cnt = 0; tot = 0
    for name, mod in mods.items():
        for k, v in mod.state_dict().items():
            got, ref, init = v.double().cpu().numpy(), g["final_%s__%s" % (name, k)], g["init_%s__%s" % (name, k)].astype(float)
            d = np.abs(got - ref); worst = max(worst, d.max()); cnt += (d > 3e-6 + 1e-4 * np.abs(ref)).sum(); tot += d.size
            rel_upd = d / np.maximum(np.abs(ref - init), 1e-12)
            if d.max() > 3e-6: print(mode, name, k, "max abs %.2e" % d.max(), "n>tol", (d > 3e-6 + 1e-4 * np.abs(ref)).sum(), "median rel-to-update %.1e" % np.median(rel_upd), "max update %.1e" % np.abs(ref - init).max())
    print(mode, "worst abs diff %.2e, %d of %d outside" % (worst, cnt, tot))
