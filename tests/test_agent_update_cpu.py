"""Host logic of the PPO update (AgentEgo.update_params) on CPU float64 against the golden run of the
reference's AgentEgo. The GAE kernel is HIP-only, so here (test infrastructure) the oracle's GAE stands in
for K5 -- K5 itself is checked on the GPU in test_hip_parity.py."""
import types

import numpy as np
import torch

from conftest import load_golden
from egopose_amd.agent import AgentEgo
from egopose_amd.nets import MLP, PolicyGaussian, Value, VideoStateNet
from egopose_amd.rl_core import TrajBatchEgo, Memory, LoggerRL
from oracle.gae import estimate_advantages as oracle_gae


def _sd(g, prefix):
    return {k[len(prefix):]: torch.as_tensor(g[k]) for k in g.files if k.startswith(prefix)}


def build_agent(g, device="cpu"):
    sdim, adim, cdim, hdim, margin, T_ep = [int(x) for x in g["dims"]]
    p_vs = VideoStateNet(cdim, hdim, margin, "lstm", None, False)
    v_vs = VideoStateNet(cdim, hdim, margin, "lstm", None, False)
    p_net = PolicyGaussian(MLP(sdim + hdim, [12, 10], "relu"), adim, log_std=-1.0, fix_std=True)
    v_net = Value(MLP(sdim + hdim, [12, 10], "relu"))
    for mod, name in [(p_vs, "p_vs"), (v_vs, "v_vs"), (p_net, "p"), (v_net, "v")]:
        mod.load_state_dict(_sd(g, "init_%s__" % name), strict=True)
    p_params = list(p_net.parameters()) + list(p_vs.parameters())
    v_params = list(v_net.parameters()) + list(v_vs.parameters())
    env = types.SimpleNamespace(cnn_feat=[g["cnn_feat0"], g["cnn_feat1"]], cfg=types.SimpleNamespace(seed=1))
    agent = AgentEgo(env=env, dtype=torch.float64, device=torch.device(device), running_state=None, custom_reward=None,
                     mean_action=False, render=False, num_threads=1, policy_net=p_net, policy_vs_net=p_vs,
                     value_net=v_net, value_vs_net=v_vs, optimizer_policy=torch.optim.Adam(p_params, lr=5e-3),
                     optimizer_value=torch.optim.Adam(v_params, lr=3e-3), opt_num_epochs=3, gamma=0.95, tau=0.95,
                     clip_epsilon=0.2, policy_grad_clip=[(p_params, 0.5)])

    def adv_fn(rewards, masks, values):
        a, r, _ = oracle_gae(rewards.cpu().numpy(), masks.cpu().numpy(), values.cpu().numpy(), agent.gamma, agent.tau)
        agent._seen = (a, r, values.cpu().numpy())
        return torch.as_tensor(a, device=rewards.device), torch.as_tensor(r, device=rewards.device)
    agent._advantages = adv_fn
    return agent, dict(p_vs=p_vs, v_vs=v_vs, p=p_net, v=v_net)


def test_update_params_matches_reference_run():
    g = load_golden("ppo_update.npz")
    torch.set_default_dtype(torch.float64)
    try:
        agent, mods = build_agent(g)
        mem = Memory()
        for i in range(g["states"].shape[0]):
            mem.push(g["states"][i], g["actions"][i], g["masks"][i], g["states"][i], g["rewards"][i], g["exps"][i], g["v_metas"][i])
        batch = TrajBatchEgo([mem])
        assert batch.states.shape == g["states"].shape and batch.v_metas.shape == g["v_metas"].shape
        agent.update_params(batch)
        a, r, v0 = agent._seen
        np.testing.assert_allclose(v0, g["values0"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(a, g["adv0"], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(r, g["ret0"], rtol=1e-11, atol=1e-12)
        for name, mod in mods.items():
            for k, v in mod.state_dict().items():
                np.testing.assert_allclose(v.numpy(), g["final_%s__%s" % (name, k)], rtol=1e-9, atol=1e-10, err_msg=name + "." + k)
    finally:
        torch.set_default_dtype(torch.float32)


def test_trajbatch_from_device_and_logger_merge():
    g = load_golden("logger_merge.npz")
    fields = [str(f) for f in g["fields"]]
    logs = []
    for row, ci in zip(g["per_worker"], g["per_worker_c_info"]):
        d = dict(zip(fields, row))
        logs.append(LoggerRL.from_totals(d["num_steps"], d["num_episodes"], d["total_reward"], d["min_episode_reward"],
                                         d["max_episode_reward"], d["total_c_reward"], d["min_c_reward"], d["max_c_reward"], ci))
    mg = LoggerRL.merge(logs)
    np.testing.assert_allclose([getattr(mg, f) for f in fields], g["merged"], rtol=1e-12)
    np.testing.assert_allclose(mg.avg_c_info, g["merged_avg_c_info"], rtol=1e-12)
    cols = dict(states=torch.zeros(4, 3), actions=torch.ones(4, 2), masks=torch.tensor([1, 0, 1, 0]), next_states=torch.zeros(4, 3),
                rewards=torch.arange(4.0), exps=torch.ones(4, dtype=torch.int64), v_metas=torch.zeros(4, 2, dtype=torch.int64))
    b = TrajBatchEgo.from_device(**cols)
    assert len(b) == 4 and b.masks.tolist() == [1, 0, 1, 0] and b.v_metas.shape == (4, 2)
    assert b.device_column("rewards") is cols["rewards"]
