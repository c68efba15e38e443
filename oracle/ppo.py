"""ORACLE (test infrastructure): one PPO update on a flat batch.

Functional torch-CPU float64 restatement of
  AgentEgo.update_params          ego_pose/core/agent_ego.py:34-57
  AgentPG.update_value            agents/agent_pg.py:19-26     (one MSE step per epoch)
  AgentPPO.update_policy          agents/agent_ppo.py:16-51    (full batch, use_mini_batch=False)
  AgentPPO.ppo_loss               agents/agent_ppo.py:58-65
  AgentPPO.clip_policy_grad       agents/agent_ppo.py:53-56
Pinned against tests/golden/ppo_update.npz.
"""
import numpy as np
import torch

from . import nets as N
from .gae import estimate_advantages


def ppo_loss(logp, logp_old, adv, eps):
    ratio = torch.exp(logp - logp_old)
    return -torch.min(ratio * adv, torch.clamp(ratio, 1.0 - eps, 1.0 + eps) * adv).mean()


def update_params(p_pol, p_pvs, p_val, p_vvs, batch, cnn_feat, *, margin, gamma, tau, clip_eps,
                  epochs, lr_policy, lr_value, grad_clip):
    """Mutates the four parameter dicts in place; returns a dict of intermediate quantities."""
    for d in (p_pol, p_pvs, p_val, p_vvs):
        for k in d:
            d[k] = d[k].clone().requires_grad_(k != "action_log_std")
    pol_leaves = [v for k, v in p_pol.items() if v.requires_grad] + list(p_pvs.values())
    val_leaves = list(p_val.values()) + list(p_vvs.values())
    opt_p = torch.optim.Adam(pol_leaves, lr=lr_policy)
    opt_v = torch.optim.Adam(val_leaves, lr=lr_value)

    states = N.as_t(batch["states"])
    actions = N.as_t(batch["actions"])
    cdim = cnn_feat[0].shape[1]
    idx, ctx = N.vsnet_train_ctx(batch["masks"], cnn_feat, batch["v_metas"], margin, cdim)
    with torch.no_grad():
        values0 = N.value(p_val, N.vsnet_train_forward(p_vvs, ctx, idx, states, margin))
    adv, ret, _ = estimate_advantages(batch["rewards"], batch["masks"], values0.numpy(), gamma, tau)
    adv_t, ret_t = N.as_t(adv), N.as_t(ret)
    with torch.no_grad():
        mean, std = N.policy_mean_std(p_pol, N.vsnet_train_forward(p_pvs, ctx, idx, states, margin))
        logp0 = N.gaussian_log_prob(mean, std, actions)
    sel = torch.as_tensor(np.nonzero(np.asarray(batch["exps"]))[0], dtype=torch.long)
    losses = []
    for _ in range(epochs):
        v_pred = N.value(p_val, N.vsnet_train_forward(p_vvs, ctx, idx, states, margin))
        v_loss = (v_pred - ret_t).pow(2).mean()
        opt_v.zero_grad()
        v_loss.backward()
        opt_v.step()
        mean, std = N.policy_mean_std(p_pol, N.vsnet_train_forward(p_pvs, ctx, idx, states, margin)[sel])
        logp = N.gaussian_log_prob(mean, std, actions[sel])
        s_loss = ppo_loss(logp, logp0[sel], adv_t[sel], clip_eps)
        opt_p.zero_grad()
        s_loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(pol_leaves, grad_clip)
        opt_p.step()
        losses.append((float(v_loss.detach()), float(s_loss.detach()), float(gnorm)))
    return dict(values0=values0.numpy(), adv=adv, ret=ret, logp0=logp0.numpy(), losses=np.array(losses))
