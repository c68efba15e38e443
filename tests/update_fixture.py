"""Shared set-up of the `AgentEgo.update_params` parity tests (CPU float64 and the GPU paths): rebuild the agent of a
golden run of the reference (tests/golden/ppo_update*.npz, written by tools/gen_golden.py / tools/gen_golden_r2.py
importing /root/reference) from the fixture's initial parameters and hyper-parameters."""
import types

import numpy as np
import torch

from egopose_amd.agent import AgentEgo
from egopose_amd.nets import MLP, PolicyGaussian, Value, VideoStateNet
from egopose_amd.rl_core import Memory, TrajBatchEgo

MODS = ("p_vs", "v_vs", "p", "v")


def _sd(g, prefix, dtype):
    return {k[len(prefix):]: torch.as_tensor(np.asarray(g[k])).to(dtype) for k in g.files if k.startswith(prefix)}


def hyper(g):
    """(lr_policy, lr_value, grad clip, epochs, gamma, tau, clip_epsilon, log_std) of the golden run."""
    if "hyper" in g.files:
        h = g["hyper"]
        return dict(lr_p=float(h[0]), lr_v=float(h[1]), clip=float(h[2]), epochs=int(h[3]), gamma=float(h[4]), tau=float(h[5]),
                    eps=float(h[6]), log_std=float(h[7]))
    return dict(lr_p=5e-3, lr_v=3e-3, clip=0.5, epochs=3, gamma=0.95, tau=0.95, eps=0.2, log_std=-1.0)     # gen_golden.py G9


def build_agent(g, device="cpu", dtype=torch.float64, fused_adam=False, net_dtype=None):
    sdim, adim, cdim, hdim, margin, T_ep = [int(x) for x in g["dims"]]
    hp = hyper(g)
    hsize = [int(g["init_p__net.affine_layers.%d.weight" % i].shape[0]) for i in range(2)]
    p_vs = VideoStateNet(cdim, hdim, margin, "lstm", None, False)
    v_vs = VideoStateNet(cdim, hdim, margin, "lstm", None, False)
    p_net = PolicyGaussian(MLP(sdim + hdim, hsize, "relu"), adim, log_std=hp["log_std"], fix_std=True)
    v_net = Value(MLP(sdim + hdim, hsize, "relu"))
    mods = dict(p_vs=p_vs, v_vs=v_vs, p=p_net, v=v_net)
    for name, mod in mods.items():
        mod.load_state_dict(_sd(g, "init_%s__" % name, dtype), strict=True)
        mod.to(dtype).to(device)
    p_params = list(p_net.parameters()) + list(p_vs.parameters())
    v_params = list(v_net.parameters()) + list(v_vs.parameters())
    env = types.SimpleNamespace(cnn_feat=[np.asarray(g["cnn_feat0"], np.float64), np.asarray(g["cnn_feat1"], np.float64)],
                                cfg=types.SimpleNamespace(seed=1))
    kw = {"fused": True} if fused_adam else {}
    agent = AgentEgo(env=env, dtype=dtype, device=torch.device(device), running_state=None, custom_reward=None,
                     mean_action=False, render=False, num_threads=1, policy_net=p_net, policy_vs_net=p_vs,
                     value_net=v_net, value_vs_net=v_vs, optimizer_policy=torch.optim.Adam(p_params, lr=hp["lr_p"], **kw),
                     optimizer_value=torch.optim.Adam(v_params, lr=hp["lr_v"], **kw), opt_num_epochs=hp["epochs"],
                     gamma=hp["gamma"], tau=hp["tau"], clip_epsilon=hp["eps"], policy_grad_clip=[(p_params, hp["clip"])],
                     net_dtype=net_dtype)
    return agent, mods


def batch_of(g):
    """The fixture's flat batch through the reference's own container path: Memory -> TrajBatchEgo."""
    mem = Memory()
    st = np.asarray(g["states"], np.float64)
    ac = np.asarray(g["actions"], np.float64)
    rw = np.asarray(g["rewards"], np.float64)
    for i in range(st.shape[0]):
        mem.push(st[i], ac[i], g["masks"][i], st[i], rw[i], g["exps"][i], g["v_metas"][i])
    return TrajBatchEgo([mem])


def check_final(mods, g, rtol, atol, max_outliers=0, outlier_atol=None):
    """Every parameter after the update against the reference's. `max_outliers` elements (whole model) may miss
    (rtol, atol) as long as they stay within `outlier_atol` (float32: Adam's first steps are +-lr whatever the gradient's
    size, so an element whose float64 gradient is ~0 can take the other sign)."""
    bad = 0
    for name, mod in mods.items():
        for k, v in mod.state_dict().items():
            got, ref = v.detach().double().cpu().numpy(), g["final_%s__%s" % (name, k)]
            miss = ~np.isclose(got, ref, rtol=rtol, atol=atol)
            if miss.any():
                if outlier_atol is None or np.abs(got - ref)[miss].max() > outlier_atol:
                    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=name + "." + k)
                bad += int(miss.sum())
    assert bad <= max_outliers, "%d parameter elements outside (rtol %g, atol %g)" % (bad, rtol, atol)
    return bad
