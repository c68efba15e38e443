"""Lockstep batched rollout: the MI355X replacement of the multi-process sampler.

The reference forks ``num_threads`` Python workers, each stepping ONE env with a batch-1 float64 policy
(/root/reference/agents/agent.py:29-111 with the AgentEgo hooks, ego_pose/core/agent_ego.py:18-32). Here every
env slot of the GPU is one such "worker": N slots advance in lockstep, and per tick

    policy (torch, MFMA GEMMs)  ->  engine: 15 x {K1 stable-PD on GPU <-> physics on host threads}
    -> K3 observation -> K6 batched ZFilter -> K2 imitation reward -> record -> in-batch resets

Semantics kept from the reference: per-worker step quota ``floor(min_batch_size / n_workers)`` checked only at
episode boundaries (episodes are never truncated), ``mask = 0`` on the last step of an episode,
``exp = 1 - mean_action``, ``v_meta = (expert_ind, start_ind)``, reward computed on the state after the step,
worker-ordered (here: slot-ordered), episode-contiguous batches, LoggerRL totals.
Deviation (documented in DESIGN.md section 5): the observation filter is updated with all slots' samples by block
merges instead of sample-by-sample inside worker 0 only.
"""
from __future__ import annotations

import gc
import math
import ctypes
import os
import time

import numpy as np
import torch

from . import _lib
from . import policy_step
from .rl_core import LoggerRL, TrajBatchEgo


class _PinnedRing:
    """Small ring of pinned int32 staging buffers for per-tick host->device flag uploads."""

    def __init__(self, rows, cols, device, slots=8):
        self.bufs = [torch.empty(rows, cols, dtype=torch.int32).pin_memory() for _ in range(slots)]
        self.events = [None] * slots
        self.device = device
        self.i = 0

    def upload(self, arr):
        k = self.i
        self.i = (self.i + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
        buf = self.bufs[k]
        buf.numpy()[...] = arr
        out = buf.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        return out


class _Uploader:
    """Non-blocking host->device uploads of small integer arrays through a ring of pinned buffers
    (``torch.as_tensor(np_array, device=...)`` from pageable memory synchronises the stream)."""

    def __init__(self, device, capacity, slots=32):
        self.device, self.capacity = device, int(capacity)
        self.bufs = [torch.empty(self.capacity, dtype=torch.int64).pin_memory() for _ in range(slots)]
        self.events = [None] * slots
        self.i = 0

    def __call__(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.int64).ravel()
        n = arr.shape[0]
        if n > self.capacity:
            return torch.as_tensor(arr, device=self.device)
        k = self.i
        self.i = (self.i + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
        buf = self.bufs[k][:n]
        buf.numpy()[...] = arr
        out = buf.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        return out


def global_budget(steps_done, cur_t, active, ids, a, b, t_eff, min_batch_size):
    """EGP_STEP_BUDGET=global, decided when the episodes of slots `ids` (all inside group [a, b), still flagged active) have
    ended: (park, rearm). park: the batch is covered by the steps collected plus what the OTHER running episodes deliver if
    they reach their end -- the slots stop. Otherwise they restart, and if even their next episodes leave a shortfall (the
    running episodes of an earlier decision ended early), `rearm` lists parked slots of the group to bring back: one per
    t_eff missing steps."""
    running = active.copy()
    running[ids] = False
    in_flight = int(np.maximum(t_eff - cur_t[running], 0).sum())
    short = int(min_batch_size) - (int(steps_done.sum()) + in_flight)
    if short <= 0:
        return True, np.zeros(0, np.int64)
    short -= len(ids) * int(t_eff)
    if short <= 0:
        return False, np.zeros(0, np.int64)
    parked = np.nonzero(~active[a:b])[0] + a
    return False, parked[:-(-short // int(t_eff))].astype(np.int64)


class LockstepRollout:

    def __init__(self, sim, policy_net, policy_vs_net, running_state=None, noise_rate=1.0, mean_action=False,
                 seed=0):
        self.sim = sim
        self.env = sim.env
        self.cfg = sim.env.cfg
        self.ctx, self.engine, self.experts = sim.ctx, sim.engine, sim.experts
        if self.experts is None:
            raise RuntimeError("LockstepRollout needs experts (env.load_experts) before sampling")
        self.policy_net, self.policy_vs_net = policy_net, policy_vs_net
        self.running_state = running_state
        self.noise_rate, self.mean_action = noise_rate, mean_action
        self.N = sim.n_env
        self.dev = torch.device("cuda", self.ctx.device)
        self.T_ep = int(self.cfg.env_episode_len)
        self.margin = int(self.cfg.fr_margin)
        # ego_forecast front end (VideoForecastNet): one video context per episode + a state LSTM stepped per tick
        self.forecast = hasattr(policy_vs_net, "s_step")
        self.ctx_T = 1 if self.forecast else self.T_ep          # context rows per episode kept in v_out
        # cfg.obs_phase (humanoid_v1.py:92-94): part of the kernel context (sim.ctx.obs_phase -- every K3 call gets the rows' cur_t);
        # cfg.random_cur_t (:218-220): an episode starts at a random step cur_t0 of its window -- state = expert frame start + cur_t0,
        # it ends when cur_t reaches the episode length. The video nets keep counting from the episode's first step, as the
        # reference's do (VideoStateNet.t / the forecast net's state LSTM restart at `initialize`), which only the torch tick knows.
        self.random_cur_t = bool(getattr(self.cfg, "random_cur_t", False))
        if getattr(self.cfg, "action_type", "position") == "torque" and hasattr(self.cfg, "j_stiff"):
            # humanoid_v1.py:56-58 writes cfg.j_stiff into the MuJoCo model's joint stiffness: a property of the physics
            # backend, which the built-in surrogate does not have
            raise NotImplementedError("cfg.j_stiff (joint stiffness under action_type 'torque') needs a physics backend that models it")
        # (the env switches of humanoid_v1.py:73-96,167-172 -- obs_heading, obs_vel, root_deheading, obs_coord (observation AND
        #  reward), action_type -- are part of the kernel context: sim.ctx.obs_dim follows them; an unknown obs_type /
        #  obs_coord / action_type was refused when it was built)
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(int(seed))
        with torch.cuda.device(self.dev):
            torch.cuda.manual_seed(int(seed) + 7919)     # default generator feeds the (graph-captured) action noise
        self.groups = [self.engine.group_range(g) for g in range(self.engine.n_groups)]
        self.rings = [_PinnedRing(4, b - a, self.dev) for a, b in self.groups]
        self.timing = {}
        self._events = [None] * len(self.groups)
        self.up = _Uploader(self.dev, max(4096, self.N))
        self.use_graphs = True              # torch tick: capture the policy step of a group in a hipGraph (eager when capture fails)
        self.use_fused = True               # HIP policy step (float32 PolicyGaussian over an MLP) wherever the nets allow it
        self.trace_ticks = False            # tools/tick_trace.py: record (group, tick, stepped envs, wait, post, reset) per env-step
        self._fused = None
        self._s_hc = None
        self._fast_bufs = None              # pinned per-tick flag / index slots of the fast tick path
        self._graphs = None                 # per group: captured hipGraph of the policy step
        self._graph_key = None
        self.reward_kind = "quat_v3"        # which entry of the reward registry the rollout evaluates (Agent sets it)
        self.custom_reward = None           # reward_kind 'callable': the caller's function, evaluated on the host per slot (env.SlotView)
        self.pool_batch = max(256, self.N // 2)
        self._pool, self._pool_pos = None, 0
        self._reset_scratch = None
        self._pool_prev, self._pool_fresh, self._ctx_keep = None, True, None
        self._prepared = None               # (key, parked generator) of a set-up made ahead by `prepare`
        self._zf_pin = None                 # pinned staging of the observation filter's state (+ the event of its last upload)

    # ------------------------------------------------------------------ helpers
    def _net_dtype(self):
        return next(self.policy_net.parameters()).dtype

    def _draw_episodes(self, n):
        """Next n (take, start frame) pairs of the reset-sampling stream + their policy video context
        (bi-LSTM over [start-m, start+T+m)), computed ahead in large batches: one fused LSTM call per
        ``pool_batch`` episodes instead of one launch-bound sweep per tick with resets."""
        out_e, out_s, out_c = [], [], []
        need = n
        while need > 0:
            if self._pool is None or self._pool_pos >= len(self._pool[0]):
                m = max(need, self.pool_batch)
                e_ind, s_ind = self.env.sample_reset(m)
                self.policy_vs_net.check_windows(e_ind, s_ind, 0 if self.forecast else self.T_ep)
                e_d, s_d = self.up(e_ind), self.up(s_ind)
                if self.forecast:        # causal net over the v_margin frames before the episode, last output
                    ctx = self.policy_vs_net.context(self.policy_vs_net.window_features(e_d, s_d)).unsqueeze(1)      # (m, 1, H)
                else:
                    win = self.policy_vs_net.window_features(e_d, s_d, self.T_ep)
                    ctx = self.policy_vs_net.forward_v_net(win)[self.margin:-self.margin].transpose(0, 1).contiguous()   # (m, T, H)
                self._pool_prev = self._pool             # (a kernel on a group stream may still be reading rows of the old pool)
                self._pool, self._pool_pos = (e_ind, s_ind, ctx), 0
                self._pool_fresh = True
            e_ind, s_ind, ctx = self._pool
            k = min(need, len(e_ind) - self._pool_pos)
            sl = slice(self._pool_pos, self._pool_pos + k)
            out_e.append(e_ind[sl]); out_s.append(s_ind[sl]); out_c.append(ctx[sl])
            self._pool_pos += k
            need -= k
        if len(out_e) == 1:
            return out_e[0], out_s[0], out_c[0]
        return np.concatenate(out_e), np.concatenate(out_s), torch.cat(out_c, 0)

    def _reset_slots(self, ids):
        """reset_model for the given slots (sorted ids): sample take/start frame, set physics state."""
        cfg, ex = self.cfg, self.experts
        e_ind, s_ind, ctx_rows = self._draw_episodes(len(ids))
        rows = ex.take_offset[e_ind] + s_ind
        t0 = self.env.np_random.randint(0, self.T_ep, size=len(ids)) if self.random_cur_t else np.zeros(len(ids), np.int64)
        qpos = ex.qpos[rows + t0].copy()             # (humanoid_v1.py:219-222: ind += cur_t)
        qvel = ex.qvel[rows + t0].copy()
        if cfg.env_init_noise > 0:
            qpos[:, 7:] += self.env.np_random.normal(0.0, cfg.env_init_noise, size=(len(ids), qpos.shape[1] - 7))
        self.engine.reset(ids, qpos, qvel)
        self.e_ind[ids], self.s_ind[ids] = e_ind, s_ind
        self.frame_base[ids] = rows
        self.cur_t[ids] = t0
        self.t0[ids] = t0
        ids_d = self.up(ids)
        self.v_out[ids_d] = ctx_rows
        if self._s_hc is not None:               # fresh episodes start the state LSTM from zero
            self._s_hc[0][ids_d] = 0
            self._s_hc[1][ids_d] = 0

    def _reset_slots_native(self, tickd_ref, g, a, b, k, ids, zf_p):
        """_reset_slots + the masked first-observation filter of slots `ids` (inside group g = [a, b)) in one library call."""
        cfg, ex = self.cfg, self.experts
        e_ind, s_ind, ctx_rows = self._draw_episodes(len(ids))
        rows = ex.take_offset[e_ind] + s_ind
        qpos = ex.qpos[rows]                         # fancy indexing copies
        qvel = ex.qvel[rows]
        if cfg.env_init_noise > 0:
            qpos[:, 7:] += self.env.np_random.normal(0.0, cfg.env_init_noise, size=(len(ids), qpos.shape[1] - 7))
        if ctx_rows.dtype != torch.float32 or not ctx_rows.is_contiguous():
            ctx_rows = ctx_rows.to(torch.float32).contiguous()
        ids32 = np.ascontiguousarray(ids, dtype=np.int32)
        e64, s64, r64 = (np.ascontiguousarray(x, dtype=np.int64) for x in (e_ind, s_ind, rows))
        if zf_p is not None:                         # same ping-pong as _obs_filter
            new_t, new, cur = self._zf_bufs[self._zf_flip], zf_p[self._zf_flip], self.zf_state.data_ptr()
            self._zf_flip ^= 1
        else:
            new_t, new, cur = None, None, None
        # (group-stream ticks read ctx_rows from another stream than the one torch made them on: the library orders itself behind
        # a fresh pool, and the rows stay referenced until the next two resets have been issued)
        self._ctx_keep = (ctx_rows, self._ctx_keep[0] if self._ctx_keep else None)
        rc = self.engine.lib.egp_rollout_reset(tickd_ref, g, a, b, k, ids32.ctypes.data, len(ids32), e64.ctypes.data, s64.ctypes.data,
                                               r64.ctypes.data, None, qpos.ctypes.data, qvel.ctypes.data, ctx_rows.data_ptr(), cur, new)
        if rc != 0:
            _lib.check(rc, "egp_rollout_reset")
        if new_t is not None:
            self.zf_state = new_t

    def _obs_filter(self, a, b, out, out2=None, active=None, write_only_active=False, phase_t=None):
        """K3+K6 fused for slots [a,b): filtered observation of the engine state -> out (and out2). obs_phase: `phase_t` = the
        slots' cur_t on the device (int32), default: uploaded from the host counters."""
        eng = self.engine
        if self.ctx.obs_phase and phase_t is None:
            phase_t = self.up(self.cur_t[a:b]).to(torch.int32)
        if self.zf_state is None:
            return self.ctx.obs_zfilter(eng.qpos[a:b], eng.qvel[a:b], None, None, 0.0, out, out2, active, write_only_active, phase_t=phase_t)
        new = self._zf_bufs[self._zf_flip]
        self._zf_flip ^= 1
        self.ctx.obs_zfilter(eng.qpos[a:b], eng.qvel[a:b], self.zf_state, new, self.zf_clip, out, out2, active, write_only_active, phase_t=phase_t)
        self.zf_state = new
        return out

    # ------------------------------------------------------------------ policy step (eager or captured in a hipGraph)
    def _mean_std(self, x):
        if hasattr(self.policy_net, "mean_std"):
            return self.policy_net.mean_std(x)
        dist = self.policy_net(x)
        return dist.loc, dist.scale

    def _policy_input(self, g, t_idx, state):
        """cat(video context of each slot's episode, state features) for group g (torch path)."""
        a, b = self.groups[g]
        ctx = self.v_out[a:b][self._ar[g], t_idx]
        st = state.to(self.v_out.dtype)
        if self.forecast:
            hc = None if self._s_hc is None else (self._s_hc[0][a:b], self._s_hc[1][a:b])
            st, hc = self.policy_vs_net.s_step(st, hc)
            if hc is not None:
                self._s_hc[0][a:b].copy_(hc[0])
                self._s_hc[1][a:b].copy_(hc[1])
        return torch.cat((ctx, st), dim=1)

    def _policy_body(self, g):
        """action = mean + std * N(0,1) for group g, reading / writing only static buffers (graph-capturable)."""
        a, b = self.groups[g]
        # the exploration noise of the tick sits in self._g_noise[g]: the caller copies it there from the rollout's noise
        # block (drawn once per rollout for every tick and slot, `sample`), so the body itself draws nothing
        if self._fused is not None:              # one HIP launch: concat + MLP + Gaussian head
            self._fused(self.v_out[a:b], self._g_tidx[g], self._g_state[g], self._g_act[g], noise=self._g_noise[g])
            return
        mean, std = self._mean_std(self._policy_input(g, self._g_tidx[g], self._g_state[g]))
        self._g_act[g].copy_(torch.addcmul(mean, std, self._g_noise[g].to(mean.dtype)))

    def _ensure_static(self, ndt):
        """Persistent buffers (and, when possible, one captured hipGraph per group) for the per-tick policy step:
        ~13 launch-bound torch ops become one graph launch."""
        # the captured graph holds raw pointers to the policy weights: moving the module (e.g. the reference's
        # `with to_cpu(...)` around checkpoint saving) re-allocates them, so the key includes their addresses
        key = (ndt, self.policy_vs_net.v_hdim) + tuple(p.data_ptr() for p in self.policy_net.parameters())
        if self.forecast:                # the captured policy step also runs the state LSTM cell of the vs net
            key += tuple(p.data_ptr() for p in self.policy_vs_net.parameters())
        if self._graph_key == key:
            if self._fused is not None:
                self._fused.refresh()            # same buffers, this iteration's weights
            return
        dev, f64 = self.dev, torch.float64
        self.v_out = torch.zeros(self.N, self.ctx_T, self.policy_vs_net.v_hdim, dtype=ndt, device=dev)
        self._s_hc = None
        if self.forecast and self.policy_vs_net.s_net_type == "lstm":
            self._s_hc = (torch.zeros(self.N, self.policy_vs_net.s_hdim, dtype=ndt, device=dev),
                          torch.zeros(self.N, self.policy_vs_net.s_hdim, dtype=ndt, device=dev))
        self._ar = [torch.arange(b - a, device=dev) for a, b in self.groups]
        self._g_tidx = [torch.zeros(b - a, dtype=torch.int64, device=dev) for a, b in self.groups]
        self._g_state = [torch.zeros(b - a, self.ctx.obs_dim, dtype=f64, device=dev) for a, b in self.groups]
        self._g_act = [torch.zeros(b - a, self.ctx.nu, dtype=f64, device=dev) for a, b in self.groups]
        self._g_noise = [torch.zeros(b - a, self.ctx.nu, dtype=torch.float32, device=dev) for a, b in self.groups]
        self._fused = None
        if self.use_fused and not self.forecast and ndt == torch.float32 and policy_step.supported(self.policy_net):
            self._fused = policy_step.FusedGaussianPolicy(self.policy_net, dev)
        self._graph_key = key
        self._graphs = None
        if not self.use_graphs:
            return
        try:
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(3):
                    for g in range(len(self.groups)):
                        self._policy_body(g)
            cur.wait_stream(side)
            graphs = []
            for g in range(len(self.groups)):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    self._policy_body(g)
                graphs.append(gr)
            self._graphs = graphs
        except Exception as e:                       # capture is an optimisation: the eager path is the same arithmetic
            self._graphs = None
            self.graph_error = repr(e)

    # ------------------------------------------------------------------ one sampling pass
    def _setup_key(self, min_batch_size, end_reward):
        """Everything a prepared set-up (prepare) depends on that the caller may change before it calls sample: the batch size, the
        end bonus, the noise regime, the reset noise, the filter object, and the weights the set-up read (the policy MLP that the
        fused step packs, the video net whose contexts fill the episode pool) by address AND version -- a checkpoint load or a
        `with to_cpu(...)` round trip in between makes the set-up stale. (The policy's log_std is read by the tick itself.)"""
        nets = [p for n, p in self.policy_net.named_parameters() if n != "action_log_std"] + list(self.policy_vs_net.parameters())
        rs = getattr(self.running_state, "rs", None)      # the set-up uploads the filter's CONTENTS and runs its first pass: an in-place
        rs_sig = None if rs is None else (int(rs._n), float(np.sum(rs._M)), float(np.sum(rs._S)))       # restore / merge makes it stale
        return (int(min_batch_size), float(end_reward), bool(self.mean_action), bool(self.noise_rate >= 1.0), float(self.cfg.env_init_noise),
                id(self.running_state), rs_sig, self.reward_kind, id(self.custom_reward), getattr(self, "step_budget", None) or os.environ.get("EGP_STEP_BUDGET", "slot"),
                getattr(self.env, "fix_len", None), getattr(self.env, "fix_head_lb", None), id(getattr(self.env, "expert_arr", None)),
                tuple((p.data_ptr(), p._version) for p in nets))

    def prepare(self, min_batch_size, end_reward=0.0):
        """Run the set-up of the NEXT sampling pass now -- record buffers, the first reset of every slot (host physics), the episode
        context pool, the first observations, the exploration noise: every launch goes onto the caller's stream BEHIND whatever is
        queued there -- and park it. AgentEgo.update_params calls this once its last epoch is enqueued, while the GPU still works
        through the update and the host would only wait: the set-up's ~7 ms of host time leave the sampling pass. `sample` takes
        the parked set-up over when nothing it depends on has changed (`_setup_key`), else drops it and starts afresh: same launches
        in the same stream order either way, so the rollout's numbers do not depend on whether it was prepared."""
        self.drop_prepared()
        key = self._setup_key(min_batch_size, end_reward)
        # the set-up draws from the device's default generator (noise block), self.gen and the env's reset stream: a set-up that is
        # dropped must leave them as it found them, or the next pass's numbers would depend on whether a prepared one was thrown away
        npr = getattr(self.env, "np_random", None)
        rng = (torch.cuda.get_rng_state(self.dev), self.gen.get_state(), None if npr is None else npr.get_state())
        gen = self._sample_gen(min_batch_size, end_reward)
        with torch.no_grad():
            next(gen)
        self._prepared = (key, gen, rng)

    def _discard(self, pr):
        pr[1].close()
        torch.cuda.set_rng_state(pr[2][0], self.dev)
        self.gen.set_state(pr[2][1])
        if pr[2][2] is not None:
            self.env.np_random.set_state(pr[2][2])

    def drop_prepared(self):
        pr, self._prepared = getattr(self, "_prepared", None), None
        if pr is not None:
            self._discard(pr)

    def sample(self, min_batch_size, end_reward=0.0):
        t_call = time.time()
        pr, self._prepared = getattr(self, "_prepared", None), None
        gen = None
        if pr is not None:
            if pr[0] == self._setup_key(min_batch_size, end_reward):
                gen = pr[1]
            else:
                self._discard(pr)
        with torch.no_grad():
            if gen is None:
                gen = self._sample_gen(min_batch_size, end_reward)
                next(gen)
            self._t_resume, self._was_prepared = t_call, pr is not None and gen is pr[1]
            if self._was_prepared and self._fused is not None:
                # the one value of the policy the set-up copied that a driver rewrites between update and sample
                # (`policy_net.action_log_std.fill_(cfg.adp_log_std)`, ego_mimic.py:101-102): re-read it now
                self._fused.log_std.copy_(self.policy_net.action_log_std.reshape(-1))
            try:
                next(gen)
            except StopIteration as done:
                return done.value
        raise RuntimeError("the sampling pass did not finish")

    def _sample_gen(self, min_batch_size, end_reward=0.0):
        """The sampling pass as a generator: set-up, `yield` (the point `prepare` parks it at), tick loop + batch assembly; the
        generator's return value is (batch, log)."""
        t_start = time.time()
        cfg, N, dev, T_ep = self.cfg, self.N, self.dev, self.T_ep
        ctx, eng = self.ctx, self.engine
        ndt = self._net_dtype()
        quota = max(1, int(math.floor(min_batch_size / N)))
        T_max = quota + T_ep
        # When does a slot whose episode has just ended stop? 'slot' (default): when IT has its share of the batch, the
        # reference's per-worker rule (agents/agent.py:36,93: every worker loops `while num_steps < thread_batch_size`).
        # 'global' (EGP_STEP_BUDGET / self.step_budget): when the batch as a whole is covered -- the same loop condition applied to all
        # slots together, counting what the running episodes can still deliver: a slot starts a new episode only while
        # (steps collected) + (steps the episodes in flight have left if they run to their end) < min_batch_size. The batch
        # still reaches min_batch_size (when the episodes in flight fall short, restarts resume and parked slots of the group
        # come back: `slot_finished`; checked before the batch is returned). With 1 024 slots and
        # 200-step episodes the per-slot rule restarts every episode that fails before its 48th step and the rollout ends
        # with ~45 ticks that step a few dozen envs; the global rule ends with the longest first episode (bench.py leg).
        budget = getattr(self, "step_budget", None) or os.environ.get("EGP_STEP_BUDGET", "slot")
        if budget not in ("slot", "global"):
            raise ValueError("step budget must be 'slot' or 'global', got %r" % (budget,))
        if budget == "global":
            T_max = quota + 3 * T_ep          # (slots may start another episode late, when the ones in flight fell short)
        H = self.policy_vs_net.v_hdim
        self.policy_vs_net.attach_feature_table(self.experts.cnn_table(dev, ndt), self.experts.cnn_offset)
        self._pool, self._pool_pos = None, 0          # contexts depend on this iteration's weights
        od, nu = ctx.obs_dim, ctx.nu
        f64 = torch.float64
        # reward_kind 'env' (custom_reward=None, agents/agent.py:56-58): the batch's reward is env_reward = 1.0 per step
        # (humanoid_v1.py:188) -- the registry's constant kernel writes exactly that -- and the logger's c_reward / c_info are 0
        reward_kernel = "constant" if self.reward_kind in ("env", "callable") else self.reward_kind
        slot_view = None
        if self.reward_kind == "callable":
            from .env import SlotView
            slot_view = SlotView(self.env)
        # time-major record in HBM. rec["states"][k] IS the policy input of tick k: the filtered observation of
        # tick k-1 is written straight into row k (and into next_states[k-1]) by the fused kernel.
        rec = dict(
            states=torch.empty(T_max + 1, N, od, dtype=f64, device=dev), next_states=torch.empty(T_max, N, od, dtype=f64, device=dev),
            actions=torch.zeros(T_max, N, nu, dtype=f64, device=dev), rewards=torch.zeros(T_max, N, dtype=f64, device=dev),
            cinfo=torch.zeros(T_max, N, ctx.reward_cinfo_dim(reward_kernel), dtype=f64, device=dev),
            exps=torch.ones(T_max, N, dtype=torch.int64, device=dev))
        host = dict(valid=np.zeros((T_max, N), bool), done=np.zeros((T_max, N), bool),
                    e_ind=np.zeros((T_max, N), np.int64), s_ind=np.zeros((T_max, N), np.int64))
        t_parts = [("alloc", time.time())]
        if self.random_cur_t:
            host["t0"] = np.zeros((T_max, N), np.int64)
        self._ensure_static(ndt)
        t_parts.append(("static", time.time()))
        self.cur_t = np.zeros(N, np.int64)
        self.t0 = np.zeros(N, np.int64)               # cur_t at the episode's first step (random_cur_t; else 0)
        self.e_ind = np.zeros(N, np.int64)
        self.s_ind = np.zeros(N, np.int64)
        self.frame_base = np.zeros(N, np.int64)
        steps_done = np.zeros(N, np.int64)
        active = np.ones(N, bool)

        def slot_finished(ids, a, b):
            """Which of the slots `ids` (group [a, b)) whose episode has just ended stop for good, and which parked slots of the
            group come back. 'global' budget: a slot parks while the steps collected plus what the running episodes can still
            deliver cover the batch; when episodes in flight end early and leave a shortfall, the slots that have just ended go
            on AND as many parked slots of this group as the shortfall needs are re-armed (they are reset with the others), so the
            remainder is not left to a handful of slots one step per tick."""
            done_ = steps_done[ids] >= quota
            rearm = np.zeros(0, np.int64)
            if budget == "global":
                t_eff = T_ep if self.env.fix_len is None else self.env.fix_len
                park, rearm = global_budget(steps_done, self.cur_t, active, ids, a, b, t_eff, min_batch_size)
                done_[:] = park
            return done_, rearm

        def after_episodes(ids, a, b):
            """Bookkeeping of the slots whose episode ended in this tick; returns the (sorted) slots to reset now."""
            finished, rearm = slot_finished(ids, a, b)
            active[ids[finished]] = False
            again = ids[~finished]
            if len(rearm):
                active[rearm] = True
                again = np.sort(np.concatenate([again, rearm]))
            return again
        if self.running_state is not None:
            rs = self.running_state.rs
            self.zf_delta_base = (float(rs._n), np.array(rs._M, float).ravel().copy(), np.array(rs._S, float).ravel().copy())
            # (through a pinned buffer: a copy from pageable memory would block the host until everything queued on the stream has
            #  run -- fatal for a set-up that `prepare` queues behind a whole update)
            st = self.running_state.to_device_state("cpu")
            if self._zf_pin is None or self._zf_pin[0].numel() != st.numel():
                self._zf_pin = [torch.empty(st.numel(), dtype=torch.float64).pin_memory(), None]
            if self._zf_pin[1] is not None:
                self._zf_pin[1].synchronize()
            self._zf_pin[0].copy_(st)
            self.zf_state = self._zf_pin[0].to(dev, non_blocking=True)
            self._zf_pin[1] = torch.cuda.Event()
            self._zf_pin[1].record()
            self._zf_bufs = [torch.empty_like(self.zf_state), torch.empty_like(self.zf_state)]
            self._zf_flip = 0
            self.zf_clip = float(self.running_state.clip or 0.0)
        else:
            self.zf_state = None
        lb = self.experts.head_height_lb
        ep_lens = []
        tick = [0] * len(self.groups)
        tm = dict(policy=0.0, wait=0.0, post=0.0, reset=0.0, small_group_ticks=0, small_group_tick_s=0.0)
        last_post = [None] * len(self.groups)

        def note_tick(g, n_stepped, now):
            """env-steps of a group that stepped fewer than 64 envs (the latency-bound tail of a rollout): how many, and
            the sum of their periods (time since the group's previous env-step ended; the groups overlap, so divide by
            their number for wall time)."""
            if last_post[g] is not None and n_stepped < 64:
                tm["small_group_ticks"] += 1
                tm["small_group_tick_s"] += now - last_post[g]
            last_post[g] = now
        trace = [] if self.trace_ticks else None

        # ---- initial reset of every slot; group g's first state goes to rec["states"][0, a:b]
        self._reset_slots(np.arange(N))
        t_parts.append(("zf+reset", time.time()))
        self._obs_filter(0, N, rec["states"][0])
        t_parts.append(("obs", time.time()))

        plain_noise = (not self.mean_action) and self.noise_rate >= 1.0
        # exploration noise of the whole rollout in ONE draw (T_max x N x nu float32, ~50 MB at the bench shape) instead of a
        # launch per group and tick on the chain filter -> policy -> env-step; both tick implementations read the same block
        noise_all = torch.randn(T_max, N, nu, dtype=torch.float32, device=dev) if plain_noise else None
        noise_p = noise_all.data_ptr() if noise_all is not None else 0

        def pre_step(g):
            a, b = self.groups[g]
            t0 = time.time()
            k = tick[g]
            t_idx = self.up(np.minimum(self.cur_t[a:b] - self.t0[a:b], self.ctx_T - 1))
            if plain_noise:
                # static-buffer form (one hipGraph launch when captured)
                self._g_tidx[g].copy_(t_idx)
                self._g_state[g].copy_(rec["states"][k, a:b])
                self._g_noise[g].copy_(noise_all[k, a:b])
                if self._graphs is not None:
                    self._graphs[g].replay()
                else:
                    self._policy_body(g)
                rec["actions"][k, a:b] = self._g_act[g]
            else:
                mean, std = self._mean_std(self._policy_input(g, t_idx, rec["states"][k, a:b]))
                if self.mean_action:
                    action = mean
                    rec["exps"][k, a:b] = 0
                else:
                    use_mean = torch.rand(b - a, device=dev, generator=self.gen) >= self.noise_rate
                    noise = torch.randn(mean.shape, dtype=mean.dtype, device=dev, generator=self.gen)
                    action = torch.where(use_mean.unsqueeze(1), mean, mean + std * noise)
                    rec["exps"][k, a:b] = (~use_mean).to(torch.int64)
                rec["actions"][k, a:b] = action          # float64 copy the engine reads in place
            ev = torch.cuda.Event()
            ev.record()
            self._events[g] = ev          # must outlive the workers' hipStreamWaitEvent
            eng.step_async(g, rec["actions"][k], active.astype(np.int32), ev)
            tm["policy"] += time.time() - t0

        def post_step(g):
            a, b = self.groups[g]
            t0 = time.time()
            eng.wait(g)
            t1 = time.time()
            k = tick[g]
            act_g = active[a:b]
            self.cur_t[a:b] += act_g
            head_z = eng.head_z[a:b]
            if self.env.fix_head_lb is not None:
                fail = head_z < self.env.fix_head_lb
            else:
                fail = head_z < lb[self.e_ind[a:b]] - 0.1
            end = self.cur_t[a:b] >= (T_ep if self.env.fix_len is None else self.env.fix_len)
            done = (fail | end) & act_g
            flags = np.stack([self.cur_t[a:b], self.frame_base[a:b] + self.cur_t[a:b], end & act_g, act_g]).astype(np.int32)
            fl = self.rings[g].upload(flags)
            host["valid"][k, a:b], host["done"][k, a:b] = act_g, done
            host["e_ind"][k, a:b], host["s_ind"][k, a:b] = self.e_ind[a:b], self.s_ind[a:b]
            if self.random_cur_t:
                host["t0"][k, a:b] = self.t0[a:b]
            # K3+K6: filtered next observation -> next_states[k] and the policy input of tick k+1;  K2: reward
            self._obs_filter(a, b, rec["next_states"][k, a:b], rec["states"][k + 1, a:b], active=fl[3], phase_t=fl[0])
            if slot_view is None:
                ctx.reward(eng.qpos[a:b], eng.prev_qpos[a:b], eng.ee_wpos[a:b], fl[0], fl[1], fl[2], end_reward, active=fl[3],
                           reward_out=rec["rewards"][k, a:b], cinfo_out=rec["cinfo"][k, a:b], kind=reward_kernel)
            else:
                # custom_reward(env, state, action, info) (agents/agent.py:53-54) for every stepped slot, on host copies
                q_h, v_h = eng.qpos.cpu().numpy()[a:b], eng.qvel.cpu().numpy()[a:b]
                pq_h = eng.prev_qpos[a:b].cpu().numpy()
                bq_h, pbq_h = ctx.body_quat(eng.qpos[a:b]).cpu().numpy(), ctx.body_quat(eng.prev_qpos[a:b]).cpu().numpy()
                ee_h = eng.ee_wpos[a:b].cpu().numpy()
                st_h, ac_h = rec["states"][k, a:b].cpu().numpy(), rec["actions"][k, a:b].cpu().numpy()
                r_h = np.zeros(b - a)
                ci_h = None
                for i in np.nonzero(act_g)[0]:
                    slot_view.load(self.cur_t[a + i], self.s_ind[a + i], self.e_ind[a + i], q_h[i], v_h[i], pq_h[i], bq_h[i], pbq_h[i], ee_h[i])
                    r_i, c_i = self.custom_reward(slot_view, st_h[i], ac_h[i], {"fail": bool(fail[i]), "end": bool(end[i])})
                    c_i = np.atleast_1d(np.asarray(c_i, float))
                    if ci_h is None:
                        ci_h = np.zeros((b - a, c_i.shape[0]))
                    r_h[i], ci_h[i] = float(r_i), c_i
                if ci_h is not None:
                    if rec["cinfo"].shape[2] != ci_h.shape[1]:          # (the callable's c_info width is only known now)
                        rec["cinfo"] = torch.zeros(T_max, N, ci_h.shape[1], dtype=f64, device=dev)
                    rec["rewards"][k, a:b] = torch.as_tensor(r_h, device=dev)
                    rec["cinfo"][k, a:b] = torch.as_tensor(ci_h, device=dev)
            steps_done[a:b] += act_g
            t2 = time.time()
            if done.any():
                ids = np.nonzero(done)[0] + a
                ep_lens.extend((self.cur_t[ids] - self.t0[ids]).tolist())     # steps taken (random_cur_t: the episode began at t0)
                again = after_episodes(ids, a, b)
                if len(again):
                    self._reset_slots(again)
                    mask = np.zeros(b - a, np.int32)
                    mask[again - a] = 1
                    # fresh episodes: their first observation goes through the filter and replaces the policy input
                    self._obs_filter(a, b, rec["states"][k + 1, a:b], active=self.up(mask).to(torch.int32), write_only_active=True)
            tick[g] = k + 1
            t3 = time.time()
            tm["wait"] += t1 - t0
            tm["post"] += t2 - t1
            tm["reset"] += t3 - t2

        # ---- native tick: the same per-tick work with no torch views, no uploads and two library calls per env-step.
        # The integer flags / context row indices of a tick live in a pinned slab (two slots per group: a slot is reused two
        # ticks later, after the env-step that was ordered behind its readers) that the policy kernel copies to its device
        # twin; every tensor argument is a precomputed address; the fused policy kernel reads rec.states[k] / writes
        # rec.actions[k] directly.
        # (mean_action: the same kernel without a noise operand writes the mean; exps = 0 as agents/agent.py:45-46)
        # (the registry's two small rewards, constant / pose_dist, the forecast nets and random_cur_t take the torch tick;
        #  EGP_FAST_TICK=0 forces it: the tests replay both forms against each other)
        fast = (self._fused is not None and (plain_noise or self.mean_action) and not self.forecast and not self.random_cur_t
                and self.reward_kind == "quat_v3" and os.environ.get("EGP_FAST_TICK", "1") != "0")
        if fast:
            if self.mean_action:
                rec["exps"].zero_()
            hnd = ctx.handle
            P = {k: v.data_ptr() for k, v in rec.items()}
            qpos_p, qvel_p, prev_p, ee_p = eng.qpos.data_ptr(), eng.qvel.data_ptr(), eng.prev_qpos.data_ptr(), eng.ee_wpos.data_ptr()
            zf_p = [b_.data_ptr() for b_ in self._zf_bufs] if self.zf_state is not None else None
            ws = ctx._workspace("zf", ctx.lib.egp_zfilter_workspace_bytes(max(b - a for a, b in self.groups), od), dev)
            ws_p = ws.data_ptr()
            v_out_p, v_stride = self.v_out.data_ptr(), self.v_out.stride(0)
            fz = self._fused
            nmax = max(b - a for a, b in self.groups)
            if self._fast_bufs is None or self._fast_bufs[2] != nmax:
                # per (group, slot) one 24*nmax-byte slab: 4 x nmax int32 flags (t | frame | end | active), then nmax int64 context rows
                shape = (len(self.groups), 2, 24 * nmax)
                self._fast_bufs = (torch.zeros(shape, dtype=torch.uint8).pin_memory(), torch.zeros(shape, dtype=torch.uint8, device=dev), nmax)
            slab_hp, slab_dp = self._fast_bufs[0].data_ptr(), self._fast_bufs[1].data_ptr()
            reward_job = eng.substeps_per_launch > 1          # K2 rides behind the resident K1 on the engine's stream
            T_eff = T_ep if self.env.fix_len is None else self.env.fix_len
            end_r = float(end_reward)
            zclip = float(self.zf_clip) if self.zf_state is not None else 0.0
            act_i32 = np.ones(N, np.int32)

        cur_stream = _lib.current_stream()           # the rollout stays on one torch stream
        ev_ring = [[torch.cuda.Event(), torch.cuda.Event()] for _ in self.groups]

        # The tick's bookkeeping and its library calls in TWO native calls (egp_rollout_tick_pre / _post, include/egopose_hip.h).
        tickd = None
        if fast:
            td = _lib.RolloutTick()
            td.ctx, td.eng, td.stream = hnd, eng.handle, cur_stream
            td.n_env, td.nmax, td.obs_dim, td.nu, td.nq, td.nv = N, nmax, od, nu, ctx.nq, ctx.nv
            td.ctx_dim, td.ctx_T, td.episode_len = H, self.ctx_T, int(T_eff)
            # A tick's `post` runs only the filter's statistics pass; the apply pass rides in the next tick's policy step (one launch
            # and its dependent round trips less on the chain filter -> policy -> env-step, ~14 us per tick), except in ticks with
            # in-batch resets and in a group's last tick (egp_rollout_tick_apply)
            defer_apply = self.zf_state is not None and nmax <= int(ctx.lib.egp_obs_zfilter_split_max_rows())
            td.defer_apply = int(defer_apply)
            td.reward_job = int(bool(reward_job))
            td.has_fix_head_lb = int(self.env.fix_head_lb is not None)
            td.fix_head_lb = float(self.env.fix_head_lb) if self.env.fix_head_lb is not None else 0.0
            td.end_reward, td.zf_clip = end_r, zclip
            lb64 = np.ascontiguousarray(lb, dtype=np.float64)
            keep = [lb64, act_i32]                        # arrays the descriptor points into
            td.cur_t, td.frame_base, td.e_ind, td.s_ind = (x.ctypes.data for x in (self.cur_t, self.frame_base, self.e_ind, self.s_ind))
            td.steps_done, td.active, td.active_i32 = steps_done.ctypes.data, active.ctypes.data, act_i32.ctypes.data
            td.head_z, td.head_lb = eng.head_z.ctypes.data, lb64.ctypes.data
            td.rec_valid, td.rec_done = host["valid"].ctypes.data, host["done"].ctypes.data
            td.rec_e_ind, td.rec_s_ind = host["e_ind"].ctypes.data, host["s_ind"].ctypes.data
            td.states, td.next_states, td.actions, td.rewards, td.cinfo = P["states"], P["next_states"], P["actions"], P["rewards"], P["cinfo"]
            td.noise = None if self.mean_action else noise_p
            td.v_out, td.v_stride = v_out_p, v_stride
            td.layers, td.n_layers, td.activation = ctypes.cast(fz.desc, ctypes.c_void_p), len(fz.layers), fz.act
            td.log_std = fz.log_std.data_ptr()
            td.slab_host, td.slab_dev = slab_hp, slab_dp
            td.qpos, td.qvel, td.prev_qpos, td.ee = qpos_p, qvel_p, prev_p, ee_p
            td.zf_workspace = ws_p
            if self._reset_scratch is None or self._reset_scratch.numel() != len(self.groups) * 6 * nmax:
                self._reset_scratch = torch.zeros(len(self.groups) * 6 * nmax, dtype=torch.int32).pin_memory()
            td.reset_scratch = self._reset_scratch.data_ptr()
            ok = (all(x.dtype == np.int64 and x.flags.c_contiguous for x in (self.cur_t, self.frame_base, self.e_ind, self.s_ind, steps_done))
                  and active.dtype == np.bool_ and eng.head_z.dtype == np.float64 and host["valid"].dtype == np.bool_ and host["done"].dtype == np.bool_
                  and host["e_ind"].dtype == np.int64 and host["s_ind"].dtype == np.int64 and rec["cinfo"].shape[2] == 5)
            if ok:
                for pair in ev_ring:                     # (a torch event gets its handle with the first record)
                    for e_ in pair:
                        e_.record()
                tickd = (td, ctypes.byref(td), keep, ctypes.c_int32(0), ctypes.c_double(0.0))

        # in-tick resets through one native call (egp_rollout_reset) instead of _reset_slots + a masked _obs_filter: the same
        # launches minus the id / mask uploads and the index_put
        native_reset = tickd is not None and self._s_hc is None and self.v_out.dtype == torch.float32 and self.v_out.is_contiguous()

        pending_apply = [None] * len(self.groups)     # per group: (zf_cur, zf_new) pointers of a filter whose apply pass is still due

        def flush_apply(g, k):                       # the apply pass of tick k's filter on its own
            pa = pending_apply[g]
            if pa is not None:
                a, b = self.groups[g]
                rc = eng.lib.egp_rollout_tick_apply(tickd[1], g, a, b, k, pa[0], pa[1])
                if rc != 0:
                    _lib.check(rc, "egp_rollout_tick_apply")
                pending_apply[g] = None

        def pre_native(g):
            a, b = self.groups[g]
            t0 = time.time()
            k = tick[g]
            ev = ev_ring[g][k & 1]
            self._events[g] = ev
            pa = pending_apply[g]
            pending_apply[g] = None
            rc = eng.lib.egp_rollout_tick_pre(tickd[1], g, a, b, k, ev.cuda_event, 1 if pa else 0, pa[0] if pa else None, pa[1] if pa else None)
            if rc != 0:
                _lib.check(rc, "egp_rollout_tick_pre")
            tm["policy"] += time.time() - t0

        def post_native(g):
            a, b = self.groups[g]
            t0 = time.time()
            k = tick[g]
            if zf_p is not None:                 # same ping-pong as _obs_filter
                new_t, new, cur = self._zf_bufs[self._zf_flip], zf_p[self._zf_flip], self.zf_state.data_ptr()
                self._zf_flip ^= 1
            else:
                new_t, new, cur = None, None, None
            n_done, wait_s = tickd[3], tickd[4]
            rc = eng.lib.egp_rollout_tick_post(tickd[1], g, a, b, k, cur, new, ctypes.byref(n_done), ctypes.byref(wait_s))
            if rc != 0:
                _lib.check(rc, "egp_rollout_tick_post")
            if new_t is not None:
                self.zf_state = new_t
                if tickd[0].defer_apply:
                    pending_apply[g] = (cur, new)
            t2 = time.time()
            if n_done.value:
                ids = np.nonzero(host["done"][k, a:b])[0] + a
                ep_lens.extend((self.cur_t[ids] - self.t0[ids]).tolist())     # steps taken (random_cur_t: the episode began at t0)
                again = after_episodes(ids, a, b)
                if len(again):
                    flush_apply(g, k)            # the resets' masked filter pass continues from the merged statistics
                if len(again) and native_reset:
                    self._reset_slots_native(tickd[1], g, a, b, k, again, zf_p)
                    self._pool_fresh = False
                elif len(again):
                    self._reset_slots(again)
                    mask = np.zeros(b - a, np.int32)
                    mask[again - a] = 1
                    self._obs_filter(a, b, rec["states"][k + 1, a:b], active=self.up(mask).to(torch.int32), write_only_active=True)
            if not active[a:b].any():
                flush_apply(g, k)                # the group's last tick: no policy step follows
            tick[g] = k + 1
            t3 = time.time()
            tm["wait"] += wait_s.value
            tm["post"] += t2 - t0 - wait_s.value
            tm["reset"] += t3 - t2
            note_tick(g, int(np.count_nonzero(host["valid"][k, a:b])), t3)
            if trace is not None:
                trace.append((g, k, int(host["valid"][k, a:b].sum()), wait_s.value, t2 - t0 - wait_s.value, t3 - t2))

        if tickd is not None:
            pre_step, post_step = pre_native, post_native
        t_parts.append(("noise+descr", time.time()))
        tm["setup_parts_ms"] = {k: round((t - (t_parts[i - 1][1] if i else t_start)) * 1e3, 2) for i, (k, t) in enumerate(t_parts)}
        tm["setup"] = time.time() - t_start          # tables, record arrays, first reset of every slot, noise (host time: launches are asynchronous)
        yield                                        # <- a prepared set-up waits here for `sample`
        t_start = self._t_resume                     # (sample_time counts from the caller's `sample` call)
        tm["setup_prepared"] = bool(self._was_prepared)
        # the tick loop is a latency chain (the Python thread hands a group its next env-step ~25 us after the last one ended): keep
        # the cyclic garbage collector out of it and let it run afterwards -- a generation-0 pass costs 50-200 us, a full one tens
        # of ms. It trims rare pauses, not the typical rollout (tools/probes/outlier_probe.py, 60 rollouts each way: mean 99.5
        # against 101.0 ms, worst 110 against 131; the medians of an alternating A/B are equal).
        gc_was_on = gc.isenabled()
        if gc_was_on:
            gc.disable()
        try:
            for g in range(len(self.groups)):
                pre_step(g)
            live = [True] * len(self.groups)
            while any(live):
                for g, (a, b) in enumerate(self.groups):
                    if not live[g]:
                        continue
                    post_step(g)
                    if active[a:b].any():
                        if tick[g] >= T_max:
                            raise RuntimeError("rollout exceeded its tick budget (quota %d + episode_len %d)" % (quota, T_ep))
                        pre_step(g)
                    else:
                        live[g] = False
        finally:
            if gc_was_on:
                gc.enable()

        # ---- episode-major batch: slot by slot, each slot's ticks in order
        t_loop_end = time.time()
        torch.cuda.synchronize(dev)              # reward launches of the last env-steps live on the engine's streams
        T_used = max(tick)
        valid = host["valid"][:T_used]                                  # (T, N)
        slot, tk = np.nonzero(valid.T)                                  # sorted by slot, then tick
        flat = torch.as_tensor(tk * N + slot, device=dev)
        pick = lambda x: x[:T_used].reshape((T_used * N,) + tuple(x.shape[2:])).index_select(0, flat)
        hpick = lambda x, dt: torch.as_tensor(x[:T_used].reshape(T_used * N)[tk * N + slot].astype(dt), device=dev)
        batch = TrajBatchEgo.from_device(
            states=pick(rec["states"]), actions=pick(rec["actions"]), masks=hpick(~host["done"], np.int64),
            next_states=pick(rec["next_states"]), rewards=pick(rec["rewards"]), exps=pick(rec["exps"]),
            v_metas=torch.stack((hpick(host["e_ind"], np.int64), hpick(host["s_ind"], np.int64)), dim=1))
        # cur_t of every batch row's episode start (random_cur_t; inspection / replay: v_meta carries only take and start frame)
        self.batch_t0 = host["t0"][:T_used].reshape(T_used * N)[tk * N + slot] if self.random_cur_t else np.zeros(len(tk), np.int64)
        r = batch.device_column("rewards")
        ci = pick(rec["cinfo"])
        stats = torch.cat([r.sum().view(1), r.min().view(1), r.max().view(1), ci.sum(0)]).cpu().numpy()
        n_steps = int(r.shape[0])
        if budget == "global" and n_steps < min_batch_size:
            raise RuntimeError("rollout ended with %d steps, fewer than min_batch_size %d" % (n_steps, min_batch_size))
        ep = np.asarray(ep_lens, float)
        if self.reward_kind == "env":          # c_reward = 0.0, c_info = [0.0] on every step (agents/agent.py:57)
            stats = np.zeros(4)
        log = LoggerRL.from_totals(n_steps, len(ep), float(n_steps), ep.min(), ep.max(), stats[0], stats[1], stats[2], stats[3:])
        if self.running_state is not None:
            self.running_state.from_device_state(self.zf_state)
        torch.cuda.synchronize(dev)
        tm["assemble"] = time.time() - t_loop_end     # episode-major gather of the record, logger totals, filter state back to the host
        log.sample_time = time.time() - t_start
        tm.update(ticks=T_used, quota=quota, step_budget=budget, policy_graph=self._graphs is not None, **eng.timing())
        self.timing = tm
        self.tick_trace = trace
        return batch, log
