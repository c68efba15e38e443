"""PPO agents on the batched rollout (drop-in for the reference's agent classes on the ego_mimic path).

  Agent      agents/agent.py:9-122       sample() -> (TrajBatch, LoggerRL); hooks; set_noise_rate
  AgentPG    agents/agent_pg.py:7-57     update_value / update_policy (A2C) / update_params
  AgentPPO   agents/agent_ppo.py:6-65    clipped-surrogate update, grad-norm clip
  AgentEgo   ego_pose/core/agent_ego.py:8-57   video-context nets, v_metas, TrajBatchEgo

Constructor keywords are the reference's. ``sample`` does not fork Python workers: it drives
``LockstepRollout`` over ``num_envs`` env slots of this rank's GPU (``num_threads`` becomes the number of
host physics threads). ``update_params`` keeps everything in HBM: values -> K5 GAE (+ global
standardisation) -> ``opt_num_epochs`` full-batch epochs; with ``torch.distributed`` initialised the flat
policy+value gradient is all-reduced once per epoch (RCCL over xGMI) so that every rank applies the
gradient of the global batch mean, exactly what the reference's single-process full batch computes.

Precision. The reference driver builds everything in float64 (ego_pose/ego_mimic.py:31-32,83-90) and hands
``dtype=torch.float64`` nets to ``AgentEgo``. The HIP LSTM / fused policy kernels and the MFMA GEMMs are float32, so
on a GPU a float64 agent keeps the caller's float64 modules as MASTER weights (state_dict, checkpoints, the caller's
optimizers step them) and computes with float32 SHADOW copies (``ShadowNets``): shadow gradients are copied up into
the masters' ``.grad`` before the all-reduce / clip / optimizer steps, the stepped masters are copied down again.
``EGP_NET_DTYPE=float64`` keeps a float64 agent on the float64 torch paths (parity runs at 1e-9).
"""
from __future__ import annotations

import copy
import math
import os
import sys
import time
import types

import numpy as np
import torch

from . import dist as D
from . import optim as O
from .rl_core import LoggerRL, TrajBatch, TrajBatchEgo
from .torch_utils import to_test, to_train


def _column(batch, name, dtype, device):
    col = batch.device_column(name) if hasattr(batch, "device_column") else None
    if col is not None and col.device == device:
        return col.to(dtype)
    return torch.from_numpy(np.asarray(getattr(batch, name))).to(dtype).to(device)


def compute_dtype(dtype, device):
    """Arithmetic type of the nets for an agent declared with `dtype` on `device` (see the module docstring)."""
    if dtype == torch.float64 and torch.device(device).type == "cuda":
        want = os.environ.get("EGP_NET_DTYPE", "float32")
        if want not in ("float32", "float64"):
            raise ValueError("EGP_NET_DTYPE must be float32 or float64, got %r" % want)
        return getattr(torch, want)
    return dtype


class ShadowNets:
    """Low-precision compute copies of the caller's (master) modules, matched parameter by parameter."""

    def __init__(self, masters, dtype):
        self.masters = dict(masters)
        self.nets = {k: copy.deepcopy(m).to(dtype) for k, m in self.masters.items()}
        self.pairs = []
        for k, m in self.masters.items():
            mp, sp = dict(m.named_parameters()), dict(self.nets[k].named_parameters())
            if list(mp) != list(sp):
                raise RuntimeError("shadow copy of %s lost parameters" % k)
            self.pairs += [(mp[n], sp[n]) for n in mp]
        self._train = [(m, s) for m, s in self.pairs if m.requires_grad]

    @torch.no_grad()
    def pull(self):
        """masters -> shadows (after an optimizer step, a checkpoint load or the driver's `action_log_std.fill_`)."""
        dst, src = [s for _, s in self.pairs], [m for m, _ in self.pairs]
        if all(m.device == s.device for m, s in self.pairs):
            torch._foreach_copy_(dst, src)
        else:                                   # masters parked on the CPU (the driver's `with to_cpu(...)`)
            for d, m in zip(dst, src):
                d.copy_(m)

    def zero_grad(self):
        for _, s in self._train:
            s.grad = None

    @torch.no_grad()
    def push_grads(self):
        """shadow gradients -> the masters' .grad (master dtype), where the exchange / clip / optimizer steps find them."""
        dst, src = [], []
        for m, s in self._train:
            if s.grad is None:
                m.grad = None
                continue
            if m.grad is None or m.grad.shape != s.grad.shape:
                m.grad = torch.empty_like(m)
            dst.append(m.grad)
            src.append(s.grad)
        if dst:
            torch._foreach_copy_(dst, src)


class Agent:

    def __init__(self, env, policy_net, value_net, dtype, device, custom_reward=None, mean_action=False,
                 render=False, running_state=None, num_threads=1, num_envs=None, num_groups=None, net_dtype=None):
        self.env, self.policy_net, self.value_net = env, policy_net, value_net
        self.dtype, self.device = dtype, device
        self.cdtype = net_dtype if net_dtype is not None else compute_dtype(dtype, device)         # what the nets compute in (float32 shadows of float64 masters)
        self.shadow = None
        self.cn = types.SimpleNamespace(policy_net=policy_net, value_net=value_net)     # the nets the kernels run
        self.custom_reward = custom_reward
        self.mean_action, self.render = mean_action, render
        self.running_state = running_state
        self.num_threads = num_threads
        self.num_envs = int(num_envs if num_envs is not None else os.environ.get("EGP_NUM_ENVS", 1024))
        self.num_groups = int(num_groups if num_groups is not None else os.environ.get("EGP_NUM_GROUPS", 2))
        self.noise_rate = 1.0
        self.traj_cls = TrajBatch
        self.logger_cls = LoggerRL
        self._rollout = None
        self._bind_compute_nets()
        if self.cdtype != self.dtype and os.environ.get("EGP_QUIET", "0") != "1":
            #  a float64 driver gets float32 arithmetic on the GPU: say so once, where the agent is built
            print("egopose_amd: agent declared %s computes in %s on %s (float64 master weights for checkpoints / optimizers; "
                  "EGP_NET_DTYPE=float64 keeps float64 arithmetic)" % (str(self.dtype).replace("torch.", ""),
                                                                     str(self.cdtype).replace("torch.", ""), self.device), file=sys.stderr)

    def _master_nets(self):
        return dict(policy_net=self.policy_net, value_net=self.value_net)

    def _bind_compute_nets(self):
        """self.cn = the modules every kernel runs: the caller's own, or float32 shadows of float64 masters."""
        masters = self._master_nets()
        if any(m is None for m in masters.values()):
            masters = {k: m for k, m in masters.items() if m is not None}        # (AgentEgo binds again with its video nets)
            self.shadow, self.cn = None, types.SimpleNamespace(**masters)
        elif self.cdtype != self.dtype:
            self.shadow = ShadowNets(masters, self.cdtype)
            self.cn = types.SimpleNamespace(**self.shadow.nets)
        else:
            self.shadow = None
            self.cn = types.SimpleNamespace(**masters)
        self.sample_modules = [self.cn.policy_net]
        self.update_modules = [self.cn.policy_net, self.cn.value_net]

    def _zero_grads(self):
        up = self._get_updater() if hasattr(self, "_updater") else None
        if up is not None:
            up.zero_grad()
            return
        if getattr(self, "_grad_sync", None) is not None:
            self._grad_sync.attach()            # gradients accumulate inside the all-reduce buffer (p.grad = its views)
        else:
            self.optimizer_value.zero_grad()
            self.optimizer_policy.zero_grad()
        if self.shadow is not None:
            self.shadow.zero_grad()

    def _grads_ready(self):
        """Called after backward: hand the gradients to the parameters the optimizers own."""
        if self.shadow is not None:
            self.shadow.push_grads()

    def _params_stepped(self):
        up = getattr(self, "_updater", None) or None
        if up is not None:
            up.rebind()          # (a caller may have moved its modules: `with to_cpu(...)` around a checkpoint)
        if self.shadow is not None:
            self.shadow.pull()

    # hooks kept for subclasses / API compatibility
    def pre_episode(self):
        return

    def pre_sample(self):
        return

    def push_memory(self, memory, state, action, mask, next_state, reward, exp):
        memory.push(state, action, mask, next_state, reward, exp)

    def trans_policy(self, states):
        return states

    def trans_value(self, states):
        return states

    def set_noise_rate(self, noise_rate):
        self.noise_rate = noise_rate

    def _video_net(self):
        raise NotImplementedError("the batched sampler is built for AgentEgo (video-context policy)")

    def _get_rollout(self):
        if self._rollout is None:
            from .rollout import LockstepRollout
            dev = torch.device(self.device)
            if dev.type != "cuda":
                raise RuntimeError("egopose_amd samples on an MI355X only: device=%s has no HIP path (no CPU fallback)" % (dev,))
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
            reward_id = getattr(self.env.cfg, "reward_id", "quat_v3")
            # custom_reward=None (agents/agent.py:53-58): the batch trains on the ENV's reward (HumanoidEnv.step returns 1.0 per
            # step, humanoid_v1.py:188) and the logger sees c_reward = 0, c_info = [0] -- kind 'env', never a silent quat_v3
            # a callable without a kernel (anything but the registry's three): the reference's plug point as it is -- evaluated on the
            # host per stepped slot through env.SlotView, kind 'callable' (slow by construction; the registry's rewards never go there)
            kind = "env" if self.custom_reward is None else getattr(self.custom_reward, "egp_kernel", "callable")
            if kind not in ("env", "quat_v3", "constant", "pose_dist", "callable") or (kind == "callable" and not callable(self.custom_reward)):
                raise NotImplementedError("custom_reward %r: unknown kernel tag %r (reward_id=%s)" % (self.custom_reward, kind, reward_id))
            n_threads = None if self.num_threads in (None, 0) else int(self.num_threads)
            sim = self.env.batched(self.num_envs, idx, n_threads=n_threads, n_groups=self.num_groups)
            seed = int(getattr(self.env.cfg, "seed", 0)) * 1000 + D.rank()
            self._rollout = LockstepRollout(sim, self.cn.policy_net, self._video_net(), self.running_state, seed=seed)
            self._rollout.reward_kind = kind
            self._rollout.custom_reward = self.custom_reward if kind == "callable" else None
        return self._rollout

    def sample(self, min_batch_size):
        t0 = time.time()
        self._params_stepped()              # shadows follow whatever the caller did to the masters since the last call
        self.pre_sample()
        ro = self._get_rollout()
        ro.noise_rate, ro.mean_action = self.noise_rate, self.mean_action
        ro.sim.ctx.set_reward_weights(self.env.cfg.reward_weights)
        with to_test(*self.sample_modules):
            per_rank = int(math.ceil(min_batch_size / D.world_size()))
            self._last_min_batch = per_rank
            batch, log = ro.sample(per_rank, end_reward=float(self.env.end_reward))
        if D.world_size() > 1:      # logger totals + filter deltas: one all-gather, merged on the device (scalars beside the update's collectives)
            log = D.merge_sampling_pass(log, self.running_state, ro.zf_delta_base if self.running_state is not None else None, self.device)
        log.sample_time = time.time() - t0
        return batch, log


class AgentPG(Agent):

    def __init__(self, gamma=0.99, tau=0.95, optimizer_policy=None, optimizer_value=None, opt_num_epochs=1,
                 value_opt_niter=1, **kwargs):
        super().__init__(**kwargs)
        self.gamma, self.tau = gamma, tau
        self.optimizer_policy, self.optimizer_value = optimizer_policy, optimizer_value
        self.opt_num_epochs, self.value_opt_niter = opt_num_epochs, value_opt_niter
        self._grad_sync = None
        self._updater = None            # optim.FlatUpdater (fused clip + Adam over flat buffers) once built; False = not eligible
        # A/B handles of the update's fused forms (the tests run both sides; all on by default, no environment switches):
        self.use_fused_optim = True     # clip + both Adam steps over flat buffers (optim.FlatUpdater) instead of the torch optimizers
        self.use_fused_loss = True      # both losses and their output gradients in one launch (optim.ppo_losses)
        self.share_train_context = True    # the value net adopts the policy net's segmented batch (VideoStateNet.adopt_train_context)
        self.reuse_first_pass = True    # the no-grad value / fixed-log-prob pass doubles as epoch 0's forward
        self.update_stats = {}

    # -- pieces shared by the A2C and PPO updates -------------------------------------------------
    def _advantages(self, rewards, masks, values):
        """K5 on device + standardisation with the GLOBAL mean / unbiased std (core/common.py:5-25). This rank's sample
        counts left in `self._pending_counts` (rows, exploration rows) ride in the same all-reduce as the advantage moments and
        come back as `self._global_counts`: one scalar exchange per update."""
        ctx = self._kernel_ctx()
        adv, ret, stats = ctx.gae(rewards.contiguous(), masks.contiguous(), values.reshape(-1).contiguous(), self.gamma, self.tau)
        stats, self._global_counts = D.merge_moments_and_counts(stats, getattr(self, "_pending_counts", None))
        ctx.gae_standardize(adv, stats)
        return adv.unsqueeze(1), ret.unsqueeze(1)

    def _advantages_with_counts(self, rewards, masks, values, counts):
        """-> (advantages, returns, global counts or None). None: `_advantages` was replaced (tests put the oracle's GAE
        there) and did not exchange the counts; update_policy then does it itself."""
        self._pending_counts, self._global_counts = counts, None
        try:
            adv, ret = self._advantages(rewards, masks, values)
        finally:
            self._pending_counts = None
        return adv, ret, self._global_counts

    def _kernel_ctx(self):
        return self._get_rollout().sim.ctx

    def _value_params(self):
        return [p for g in self.optimizer_value.param_groups for p in g["params"]]

    def _policy_params(self):
        return [p for g in self.optimizer_policy.param_groups for p in g["params"]]

    def _sync_grads(self):
        if D.world_size() > 1:
            if self._grad_sync is None:
                self._grad_sync = D.FlatGradSync(self._value_params() + self._policy_params())
            self._grad_sync.all_reduce()

    def _clip_list(self):
        return []

    def _get_updater(self):
        """optim.FlatUpdater for this agent's two optimizers, or None (CPU device, non-Adam optimizers, use_fused_optim = False, ...)."""
        if self._updater is None:
            up = None
            if torch.device(self.device).type == "cuda" and self.use_fused_optim and \
                    self.optimizer_value is not None and self.optimizer_policy is not None:
                compute_of = {m: c for m, c in self.shadow.pairs} if self.shadow is not None else None
                up = O.FlatUpdater.build([self.optimizer_value, self.optimizer_policy], self._clip_list(), compute_of)
            self._updater = up if up is not None else False
        return self._updater or None

    def _optim_step(self, which=(0, 1)):
        """After backward: gradient exchange, clip, optimizer steps (0 = value, 1 = policy; the reference steps the critic
        first, agents/agent_ppo.py:24-30), compute copies refreshed."""
        up = self._get_updater()
        if up is not None:
            up.collect_grads(which)
            up.all_reduce(which)
            up.step(which)
            return
        self._grads_ready()
        self._sync_grads()
        if 0 in which:
            self.optimizer_value.step()
        if 1 in which:
            self.clip_policy_grad()
            self.optimizer_policy.step()
        self._params_stepped()

    def clip_policy_grad(self):
        return

    def _value_backward(self, states, returns, n_global):
        """MSE critic loss of the global batch (this rank's share) -> gradients, no optimizer step."""
        if self.value_opt_niter != 1:
            raise NotImplementedError("value_opt_niter != 1 is not on the ego_mimic path")
        pred = self.cn.value_net(self.trans_value(states))
        loss = (pred - returns).pow(2).sum() / n_global
        loss.backward()
        return loss

    def update_value(self, states, returns):
        """update critic (agents/agent_pg.py:19-26)"""
        self._zero_grads()
        loss = self._value_backward(states, returns, D.global_count(states.shape[0], states.device))
        self._optim_step(which=(0,))
        return loss

    def update_policy(self, states, actions, returns, advantages, exps):
        ind = exps.nonzero().squeeze(1)
        n_val = D.global_count(states.shape[0], states.device)
        n_exp = D.global_count(ind.shape[0], states.device)
        for _ in range(self.opt_num_epochs):
            self._zero_grads()
            self._value_backward(states, returns, n_val)
            logp = self.cn.policy_net.get_log_prob(self.trans_policy(states)[ind], actions[ind])
            loss = -(logp * advantages[ind]).sum() / n_exp
            loss.backward()
            self._optim_step()

    def _load_batch(self, batch):
        dev = torch.device(self.device)
        cols = {k: _column(batch, k, self.cdtype, dev) for k in ("states", "actions", "rewards", "masks", "exps")}
        return cols

    def update_params(self, batch):
        t0 = time.time()
        to_train(*self.update_modules)
        self._params_stepped()
        c = self._load_batch(batch)
        with to_test(*self.update_modules):
            with torch.no_grad():
                values = self.cn.value_net(self.trans_value(c["states"]))
        advantages, returns = self._advantages(c["rewards"], c["masks"], values)
        self.update_policy(c["states"], c["actions"], returns, advantages, c["exps"])
        return time.time() - t0


class AgentPPO(AgentPG):

    def __init__(self, clip_epsilon=0.2, opt_batch_size=64, use_mini_batch=False, policy_grad_clip=None, **kwargs):
        super().__init__(**kwargs)
        self.clip_epsilon, self.opt_batch_size = clip_epsilon, opt_batch_size
        self.use_mini_batch, self.policy_grad_clip = use_mini_batch, policy_grad_clip

    def _group_contexts(self, states=None):
        """Hook: prepare the critic's and the actor's state transforms together (AgentEgo). False = nothing prepared."""
        return False

    def _clip_list(self):
        return [(list(params), max_norm) for params, max_norm in (self.policy_grad_clip or [])]

    def clip_policy_grad(self):
        if self.policy_grad_clip is not None:
            for params, max_norm in self.policy_grad_clip:
                torch.nn.utils.clip_grad_norm_(params, max_norm)

    def _fused_losses(self):
        """Both losses and their gradients w.r.t. the nets' outputs from ONE HIP launch (optim.ppo_losses) instead of ~35
        element-wise library kernels per epoch: float32 compute nets on the GPU with a Gaussian policy head."""
        pol = self.cn.policy_net
        ls = getattr(pol, "action_log_std", None)
        return (torch.device(self.device).type == "cuda" and self.cdtype == torch.float32 and self.value_opt_niter == 1
                and hasattr(pol, "mean_std") and ls is not None and ls.dim() == 2 and ls.shape[0] == 1 and ls.shape[1] <= 256
                and ls.dtype == torch.float32 and self.use_fused_loss)

    def ppo_loss(self, states, actions, advantages, fixed_log_probs, ind, n_exp=None):
        """`ind` = rows with exps == 1 (agents/agent_ppo.py:45-51), or None when that is every row (no gather copies)."""
        if ind is None:
            n_exp = states.shape[0] if n_exp is None else n_exp
            logp = self.cn.policy_net.get_log_prob(self.trans_policy(states), actions)
            ratio = torch.exp(logp - fixed_log_probs)
            adv = advantages
        else:
            n_exp = ind.shape[0] if n_exp is None else n_exp
            logp = self.cn.policy_net.get_log_prob(self.trans_policy(states)[ind], actions[ind])
            ratio = torch.exp(logp - fixed_log_probs[ind])
            adv = advantages[ind]
        clipped = torch.clamp(ratio, 1.0 - self.clip_epsilon, 1.0 + self.clip_epsilon) * adv
        return -torch.min(ratio * adv, clipped).sum() / n_exp

    def _exploration_rows(self, exps, n):
        """Rows with exps == 1 (agents/agent_ppo.py:45-46), or None when that is every row (no gather copies)."""
        ind = exps.nonzero().squeeze(1)
        return (None if ind.shape[0] == n else ind), ind.shape[0]

    def _surrogate(self, logp, adv, fixed, n_exp):
        ratio = torch.exp(logp - fixed)
        clipped = torch.clamp(ratio, 1.0 - self.clip_epsilon, 1.0 + self.clip_epsilon) * adv
        return -torch.min(ratio * adv, clipped).sum() / n_exp

    def update_policy(self, states, actions, returns, advantages, exps, first_pass=None, counts=None):
        """`first_pass` = (pred, head, ind) of a forward pass ALREADY made with the current weights and autograd on
        (AgentEgo.update_params): it served as the no-grad value / fixed-log-prob pass and now is epoch 0's forward --
        the weights have not moved in between, so the numbers are the reference's, with two forward passes fewer.
        `head` is the policy's action mean when the fused loss kernel runs (`_fused_losses`), else its log-probabilities.
        `counts` = (global rows, global exploration rows) when the caller has exchanged them already."""
        if self.use_mini_batch:
            raise NotImplementedError("mini-batch PPO is not on the ego_mimic path (AgentEgo forces full batch)")
        fused = self._fused_losses()
        if first_pass is None:
            ind, n_ind = self._exploration_rows(exps, states.shape[0])
        else:
            ind = first_pass[2]
            n_ind = states.shape[0] if ind is None else ind.shape[0]
        if counts is None:
            n_val = D.global_count(states.shape[0], states.device)
            n_exp = D.global_count(n_ind, states.device)
        else:
            n_val, n_exp = counts
        if fused:
            return self._update_policy_fused(states, actions, returns, advantages, ind, n_ind, n_val, n_exp, first_pass)
        if first_pass is None:
            with to_test(*self.update_modules):
                with torch.no_grad():
                    fixed_log_probs = self.cn.policy_net.get_log_prob(self.trans_policy(states), actions)
        else:
            pred0, logp0, _ = first_pass
            fixed_sel = logp0.detach()
            fixed_log_probs = None
        losses = []
        for epoch in range(self.opt_num_epochs):
            # critic and actor have disjoint parameters: both backward passes run before the single gradient
            # exchange; the value step precedes the policy step as in the reference. (Running the two passes on two
            # HIP streams was faster and hung the GPU intermittently -- concurrent library GEMMs, docs/DESIGN_TRAIL.md section 2.)
            if first_pass is not None and epoch == 0:
                v_loss = (pred0 - returns).pow(2).sum() / n_val
                adv = advantages if ind is None else advantages[ind]
                s_loss = self._surrogate(logp0, adv, fixed_sel, n_exp)
                self._zero_grads()
                (v_loss + s_loss).backward()
            elif first_pass is not None:
                self._group_contexts(states)
                pred = self.cn.value_net(self.trans_value(states))
                v_loss = (pred - returns).pow(2).sum() / n_val
                x = self.trans_policy(states)
                logp = self.cn.policy_net.get_log_prob(x if ind is None else x[ind], actions if ind is None else actions[ind])
                s_loss = self._surrogate(logp, advantages if ind is None else advantages[ind], fixed_sel, n_exp)
                self._zero_grads()
                (v_loss + s_loss).backward()
            elif self._group_contexts(states):
                # both video nets' recurrences in one grouped launch each way; disjoint parameters, so one backward
                # over the sum of the two losses yields exactly the two separate gradients
                if self.value_opt_niter != 1:
                    raise NotImplementedError("value_opt_niter != 1 is not on the ego_mimic path")
                pred = self.cn.value_net(self.trans_value(states))
                v_loss = (pred - returns).pow(2).sum() / n_val
                s_loss = self.ppo_loss(states, actions, advantages, fixed_log_probs, ind, n_exp)
                self._zero_grads()
                (v_loss + s_loss).backward()
            else:
                self._zero_grads()
                v_loss = self._value_backward(states, returns, n_val)
                s_loss = self.ppo_loss(states, actions, advantages, fixed_log_probs, ind, n_exp)
                s_loss.backward()
            self._optim_step()
            losses.append((v_loss.detach(), s_loss.detach()))
        self.update_stats = {"value_loss": [float(v) for v, _ in losses], "surr_loss": [float(s) for _, s in losses]}

    def _policy_mean(self, x):
        return self.cn.policy_net.mean_std(x)[0]

    def _epochs_enqueued(self):
        """Hook: every launch of the update is on the stream, its results have not been read yet."""
        return

    def _update_policy_fused(self, states, actions, returns, advantages, ind, n_ind, n_val, n_exp, first_pass):
        """The epochs with optim.ppo_losses: forward passes -> ONE launch for both losses and d loss / d (values, action mean)
        -> autograd from those two tensors -> fused exchange / clip / Adam (`_optim_step`). Epoch 0's forward doubles as the
        pass that fixes the sampling policy's log-probabilities (the weights have not moved since the rollout), whether it
        arrives as `first_pass` or is made here."""
        dev = states.device
        n, A = states.shape[0], actions.shape[1]
        pol = self.cn.policy_net
        log_std = pol.action_log_std
        learn_std = bool(log_std.requires_grad)
        fixed = torch.empty(n_ind, dtype=torch.float32, device=dev)
        d_pred = torch.empty(n, 1, dtype=torch.float32, device=dev)
        d_mean = torch.empty(n_ind, A, dtype=torch.float32, device=dev)
        rec = torch.zeros(max(1, self.opt_num_epochs), 2, dtype=torch.float64, device=dev)
        act = actions if actions.stride(1) == 1 else actions.contiguous()
        for epoch in range(self.opt_num_epochs):
            if first_pass is not None and epoch == 0:
                pred, mean = first_pass[0], first_pass[1]
            else:
                self._group_contexts(states)
                pred = self.cn.value_net(self.trans_value(states))
                x = self.trans_policy(states)
                mean = self._policy_mean(x if ind is None else x[ind])
            self._zero_grads()
            _, _, _, d_ls = O.ppo_losses(pred.detach(), returns, mean.detach(), act, log_std.detach(), advantages, fixed, epoch == 0,
                                         self.clip_epsilon, n_val, n_exp, rows=ind, d_pred=d_pred, d_mean=d_mean,
                                         want_d_log_std=learn_std, losses_out=rec[epoch])
            torch.autograd.backward([pred, mean], [d_pred, d_mean])
            if learn_std:
                log_std.grad = d_ls.view_as(log_std)
            self._optim_step()
        self._epochs_enqueued()              # (the host would only wait from here on: the GPU still works through the epochs)
        host = rec.tolist()
        self.update_stats = {"value_loss": [r[0] for r in host[:self.opt_num_epochs]], "surr_loss": [r[1] for r in host[:self.opt_num_epochs]]}


def _quat_v3_marker(fn):
    fn.egp_kernel = "quat_v3"
    return fn


class AgentEgo(AgentPPO):

    def __init__(self, policy_vs_net=None, value_vs_net=None, **kwargs):
        super().__init__(use_mini_batch=False, **kwargs)
        self.traj_cls = TrajBatchEgo
        self.policy_vs_net, self.value_vs_net = policy_vs_net, value_vs_net
        self._bind_compute_nets()

    def _master_nets(self):
        return dict(policy_net=self.policy_net, value_net=self.value_net, policy_vs_net=getattr(self, "policy_vs_net", None),
                    value_vs_net=getattr(self, "value_vs_net", None))

    def _bind_compute_nets(self):
        super()._bind_compute_nets()
        if getattr(self.cn, "policy_vs_net", None) is not None:
            self.sample_modules = [self.cn.policy_net, self.cn.policy_vs_net]
            self.update_modules = [self.cn.policy_net, self.cn.value_net, self.cn.policy_vs_net, self.cn.value_vs_net]
            from .nets import VideoStateNet
            # When a head is an MLP on the HIP GEMMs its first layer can gather the context rows and append the state columns
            # itself: the video net is then asked (per call, `lazy_width`) for the parts instead of the concatenation. (That a
            # VideoStateNet output's state columns need no gradient travels as a tag on the tensor it returns.) Nothing is
            # stored on the caller's modules.
            self._lazy_width = {}
            for key, head, vs in (("policy", self.cn.policy_net, self.cn.policy_vs_net), ("value", self.cn.value_net, self.cn.value_vs_net)):
                layers = getattr(getattr(head, "net", None), "affine_layers", None)
                self._lazy_width[key] = int(layers[0].out_features) if (isinstance(vs, VideoStateNet) and layers is not None and len(layers) > 0) else 0

    def _video_net(self):
        return self.cn.policy_vs_net

    prefetch_rollout = True     # set up the next sampling pass behind the update's last epoch (LockstepRollout.prepare)

    def _epochs_enqueued(self):
        """The reference's loop is sample -> update -> sample (ego_pose/ego_mimic.py:106-118): once the update's launches are
        queued, the set-up of the next sampling pass -- host work the GPU does not wait for -- runs while the GPU finishes the
        update. Only with compute nets that ARE the caller's modules (shadow copies are refreshed at `sample`, which would make
        the set-up stale every time) and only when a sampling pass has told us its batch size."""
        ro, mb = self._rollout, getattr(self, "_last_min_batch", None)
        if not self.prefetch_rollout or ro is None or mb is None or self.shadow is not None or D.world_size() > 1:
            return
        with to_test(*self.sample_modules):
            self.pre_sample()
            ro.noise_rate, ro.mean_action = self.noise_rate, self.mean_action
            ro.prepare(mb, end_reward=float(self.env.end_reward))

    def pre_sample(self):
        # ego_pose/core/agent_ego.py:18-19: the video net leaves the update in 'train' mode; every sampling pass starts in 'test'
        self.cn.policy_vs_net.set_mode("test")

    def pre_episode(self):
        self.cn.policy_vs_net.initialize(torch.as_tensor(self.env.get_episode_cnn_feat()))

    def push_memory(self, memory, state, action, mask, next_state, reward, exp):
        memory.push(state, action, mask, next_state, reward, exp, np.array([self.env.expert_ind, self.env.start_ind]))

    def _group_contexts(self, states=None):
        from .nets import grouped_forecast_context, grouped_video_context
        nets = [self.cn.value_vs_net, self.cn.policy_vs_net]
        return grouped_video_context(nets) or grouped_forecast_context(nets, states)

    def trans_policy(self, states):
        w = getattr(self, "_lazy_width", {}).get("policy", 0)
        return self.cn.policy_vs_net(states, lazy_width=w) if w else self.cn.policy_vs_net(states)

    def trans_value(self, states):
        w = getattr(self, "_lazy_width", {}).get("value", 0)
        return self.cn.value_vs_net(states, lazy_width=w) if w else self.cn.value_vs_net(states)

    def update_params(self, batch):
        t0 = time.time()
        to_train(*self.update_modules)
        self._params_stepped()
        c = self._load_batch(batch)
        dev = c["states"].device
        vs_nets = (self.cn.policy_vs_net, self.cn.value_vs_net)
        v_metas = batch.device_column("v_metas") if hasattr(batch, "device_column") else None
        v_metas = v_metas.cpu().numpy() if v_metas is not None else batch.v_metas
        if self._rollout is not None and self._rollout.experts is not None:
            ex = self._rollout.experts
            pdt = next(vs_nets[0].parameters()).dtype
            for net in vs_nets:
                net.attach_feature_table(ex.cnn_table(dev, pdt), ex.cnn_offset)
        x_init = (c["masks"], self.env.cnn_feat, v_metas)
        for i, net in enumerate(vs_nets):
            net.set_mode("train")
            if i > 0 and hasattr(net, "adopt_train_context") and self.share_train_context:
                net.adopt_train_context(vs_nets[0], x_init)      # same batch, same feature table: segment and gather once
            else:
                net.initialize(x_init)
        n_rows = c["states"].shape[0]
        ind, n_ind = self._exploration_rows(c["exps"], n_rows)
        if self.value_opt_niter == 1 and self.reuse_first_pass:
            # ONE forward pass with autograd on serves three purposes: the values that GAE consumes, the fixed log-probs of
            # the surrogate, and epoch 0's forward (nothing has stepped in between; no dropout / batch norm in these nets)
            self._group_contexts(c["states"])
            pred0 = self.cn.value_net(self.trans_value(c["states"]))
            x = self.trans_policy(c["states"])
            xs = x if ind is None else x[ind]
            if self._fused_losses():
                head0 = self._policy_mean(xs)           # the loss kernel forms the log-probabilities itself
            else:
                head0 = self.cn.policy_net.get_log_prob(xs, c["actions"] if ind is None else c["actions"][ind])
            advantages, returns, counts = self._advantages_with_counts(c["rewards"], c["masks"], pred0.detach(), (n_rows, n_ind))
            self.update_policy(c["states"], c["actions"], returns, advantages, c["exps"], first_pass=(pred0, head0, ind), counts=counts)
        else:
            with to_test(*self.update_modules):
                with torch.no_grad():
                    self._group_contexts(c["states"])       # the policy net's context is consumed by update_policy's first pass
                    values = self.cn.value_net(self.trans_value(c["states"]))
            advantages, returns, counts = self._advantages_with_counts(c["rewards"], c["masks"], values, (n_rows, n_ind))
            self.update_policy(c["states"], c["actions"], returns, advantages, c["exps"], counts=counts)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        return time.time() - t0
