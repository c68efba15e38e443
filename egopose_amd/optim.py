"""The element-wise tail of a PPO epoch on the HIP kernels of csrc/egp_update.hip.

  ppo_losses(...)   critic MSE (agents/agent_pg.py:19-26) + clipped surrogate (agents/agent_ppo.py:58-65 over
                    core/distributions.py:6-25) and their gradients w.r.t. `values_pred` / `action_mean` in ONE launch; the caller
                    back-propagates from those two tensors (no scalar-loss graph)
  FlatUpdater       clip_policy_grad (agent_ppo.py:53-56) + optimizer_value.step() + optimizer_policy.step() (agent_ppo.py:24-30)
                    as two launches over flat buffers. The caller's torch.optim.Adam objects stay the source of every
                    hyper-parameter (`set_optimizer_lr` keeps working) and see the moments as their own state; the parameters of
                    the caller's modules become views of one flat buffer, the gradient buffer doubles as the all-reduce buffer
                    (no copies in or out), and with float64 master modules (the unmodified driver) the float32 compute copies
                    are refreshed by the same launch.

Neither has a CPU form: on a CPU device (the oracle-side parity tests) the agent runs the plain torch formulation.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from . import dist as D

_WS = {}            # (key, device, stream handle) -> zeroed scratch; least recently used entries dropped beyond _WS_MAX
_WS_MAX = 8


def _zeroed_workspace(key, nbytes, device):
    """Partial-sum scratch of the loss kernels, one per (device, stream): two updates issued on different streams of one device
    must not share it between k_ppo_loss and k_ppo_loss_final. INVARIANT: the buffer is all zeros between calls --
    the kernels' arrival counter lives in it, starts at zero and is reset by the last workgroup to arrive; a fresh buffer is
    zero-filled here. The cache is bounded (code that creates short-lived streams must not grow it for ever): an evicted entry's
    stream handle may be recycled by the runtime for a new stream, which then simply gets a fresh zeroed buffer."""
    k = (key, str(device), int(torch.cuda.current_stream(device).cuda_stream))
    buf = _WS.pop(k, None)
    if buf is None or buf.numel() < nbytes:
        buf = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
    _WS[k] = buf                       # (re-inserted: dict order = recency)
    while len(_WS) > _WS_MAX:
        _WS.pop(next(iter(_WS)))
    return buf


def losses_available(pred, mean, log_std):
    return (pred.is_cuda and pred.dtype == torch.float32 and mean.dtype == torch.float32 and mean.dim() == 2 and mean.stride(1) == 1
            and log_std.dtype == torch.float32 and log_std.numel() == mean.shape[1] and mean.shape[1] <= 256)


def ppo_losses(pred, returns, mean, actions, log_std, adv, fixed_logp, write_fixed, clip_eps, n_val, n_exp, rows=None,
               d_pred=None, d_mean=None, want_d_log_std=False, losses_out=None):
    """-> (losses float64[2] on the device = value loss, surrogate loss; d_pred (n, 1); d_mean (n_pol, A); d_log_std (1, A) or None).
    `fixed_logp` (n_pol,) float32 is written when `write_fixed`, read otherwise. `rows`: int64 sample index of every policy row."""
    lib = L.load()
    n, n_pol, A = pred.shape[0], mean.shape[0], mean.shape[1]
    dev = pred.device
    pred1 = pred.reshape(-1)
    ret1 = returns.reshape(-1)
    adv1 = adv.reshape(-1)
    for name, t, want in (("pred", pred1, n), ("returns", ret1, n), ("adv", adv1, n), ("fixed_logp", fixed_logp, n_pol)):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == want):
            raise ValueError("%s must be a contiguous float32 HIP tensor of %d elements" % (name, want))
    if not (actions.dtype == torch.float32 and actions.dim() == 2 and actions.stride(1) == 1 and actions.shape == (n, A)):
        raise ValueError("actions must be (n, act_dim) float32 with unit column stride")
    ls = log_std.reshape(-1).contiguous()
    if rows is not None and not (rows.dtype == torch.int64 and rows.is_contiguous() and rows.numel() == n_pol):
        raise ValueError("rows must be a contiguous int64 tensor with one entry per policy row")
    if d_pred is None:
        d_pred = torch.empty(n, 1, dtype=torch.float32, device=dev)
    if d_mean is None:
        d_mean = torch.empty(n_pol, A, dtype=torch.float32, device=dev)
    d_ls = torch.empty(1, A, dtype=torch.float32, device=dev) if want_d_log_std else None
    losses = losses_out if losses_out is not None else torch.empty(2, dtype=torch.float64, device=dev)
    ws = _zeroed_workspace("ppo_loss", lib.egp_ppo_loss_workspace_bytes(n, n_pol, A), dev)
    d = L.PpoLossDesc()
    d.n, d.n_pol, d.act_dim = n, n_pol, A
    d.rows = rows.data_ptr() if rows is not None else None
    d.pred, d.returns = pred1.data_ptr(), ret1.data_ptr()
    d.mean, d.ld_mean = mean.data_ptr(), mean.stride(0) if n_pol > 1 else A
    d.actions, d.ld_act = actions.data_ptr(), actions.stride(0) if n > 1 else A
    d.log_std, d.adv = ls.data_ptr(), adv1.data_ptr()
    d.fixed_logp, d.write_fixed = fixed_logp.data_ptr(), 1 if write_fixed else 0
    d.clip_eps, d.inv_n_val, d.inv_n_exp = float(clip_eps), 1.0 / max(float(n_val), 1.0), 1.0 / max(float(n_exp), 1.0)
    d.d_pred = d_pred.data_ptr()
    d.d_mean, d.ld_dmean = d_mean.data_ptr(), d_mean.stride(0) if n_pol > 1 else A
    d.d_log_std = d_ls.data_ptr() if d_ls is not None else None
    d.losses, d.workspace = losses.data_ptr(), ws.data_ptr()
    L.check(lib.egp_ppo_loss_f32(C.byref(d), L.current_stream()), "egp_ppo_loss_f32")
    return losses, d_pred, d_mean, d_ls


class FlatUpdater:
    """Flat storage + fused clip / Adam step for the parameter groups of `optimizers` (stepped in that order, all at once).

    `clip`: AgentPPO.policy_grad_clip -- [(params, max_norm), ...]. `compute_of`: {master parameter: float32 compute copy} when
    the optimizers own float64 master modules (agent.ShadowNets), else None (the optimizers' parameters compute themselves)."""

    @staticmethod
    def build(optimizers, clip, compute_of=None):
        """A FlatUpdater, or None when this set-up is not the one the kernels implement (the caller then keeps torch's path)."""
        try:
            return FlatUpdater(optimizers, clip, compute_of)
        except _NotEligible:
            return None

    def __init__(self, optimizers, clip, compute_of=None):
        self.optimizers = list(optimizers)
        self.lib = L.load()
        clip = list(clip or [])
        clip_of = {}
        for gi, (params, max_norm) in enumerate(clip, start=1):
            for p in params:
                if id(p) in clip_of:
                    raise _NotEligible("a parameter sits in two clip lists")
                clip_of[id(p)] = (gi, float(max_norm))
        self.entries = []            # (optimizer index, group index, master param, clip group, max_norm)
        for oi, opt in enumerate(self.optimizers):
            if type(opt) is not torch.optim.Adam:
                raise _NotEligible("not torch.optim.Adam")
            for gi, g in enumerate(opt.param_groups):
                if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or g.get("capturable"):
                    raise _NotEligible("Adam variant")
                for p in g["params"]:
                    if p.requires_grad:
                        cg, mn = clip_of.get(id(p), (0, 0.0))
                        self.entries.append((oi, gi, p, cg, mn))
        if not self.entries:
            raise _NotEligible("nothing to train")
        masters = [e[2] for e in self.entries]
        dev, mdt = masters[0].device, masters[0].dtype
        if dev.type != "cuda" or mdt not in (torch.float32, torch.float64) or any(p.device != dev or p.dtype != mdt for p in masters):
            raise _NotEligible("parameters must be float32 / float64 on one HIP device")
        if len({id(p) for p in masters}) != len(masters):
            raise _NotEligible("a parameter is owned by two groups")
        self.compute = [compute_of[p] if compute_of is not None else p for p in masters]
        cdt = self.compute[0].dtype
        if any(c.dtype != cdt or c.device != dev or c.shape != p.shape for c, p in zip(self.compute, masters)):
            raise _NotEligible("compute copies do not match their masters")
        if (mdt, cdt) not in ((torch.float32, torch.float32), (torch.float64, torch.float32), (torch.float64, torch.float64)):
            raise _NotEligible("unsupported dtype pair")
        if compute_of is None and cdt != mdt:
            raise _NotEligible("dtype pair without compute copies")
        self.device, self.mdt, self.cdt = dev, mdt, cdt
        self.shadowed = compute_of is not None
        # segments: maximal runs of consecutive parameters with the same (optimizer, group, clip group)
        self.offsets, pos = [], 0
        self.segments = []           # [begin, end, optimizer index, group index, clip group, max_norm]
        for oi, gi, p, cg, mn in self.entries:
            self.offsets.append(pos)
            if self.segments and self.segments[-1][2:5] == [oi, gi, cg]:
                self.segments[-1][1] = pos + p.numel()
            else:
                self.segments.append([pos, pos + p.numel(), oi, gi, cg, mn])
            pos += p.numel()
        self.numel = pos
        if len(self.segments) > L.ADAM_MAX_SEGMENTS:
            raise _NotEligible("more parameter segments than the kernel takes")
        mk = lambda dt: torch.zeros(self.numel, dtype=dt, device=dev)
        self.P, self.M, self.V = mk(mdt), mk(mdt), mk(mdt)
        self.G = mk(cdt)                                         # gradients of the compute parameters = the all-reduce buffer
        self.S = mk(torch.float32) if self.shadowed else None    # float32 compute copies of float64 masters
        self.steps = [0] * len(self.optimizers)
        self.time_collectives, self.collective_events = False, []
        # what the optimizers' state shows as `step`: a tensor of its OWN per parameter (torch's optimizer.step() increments every
        # state["step"] it finds -- one shared tensor per optimizer would be incremented once per parameter)
        self.step_tensors = [torch.tensor(0.0) for _ in self.entries]
        self._views = lambda flat: [flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, masters)]
        self.p_views, self.m_views, self.v_views, self.g_views = (self._views(f) for f in (self.P, self.M, self.V, self.G))
        self.s_views = self._views(self.S) if self.shadowed else None
        self.ws = torch.empty(int(self.lib.egp_adam_workspace_bytes()), dtype=torch.uint8, device=dev)
        self.norms = torch.zeros(L.ADAM_MAX_SEGMENTS + 1, dtype=torch.float64, device=dev)
        # moments the optimizers may already hold (a resumed run) move into the flat buffers; from now on the optimizers' state
        # entries ARE views of them
        with torch.no_grad():
            for k, (oi, gi, p, _, _) in enumerate(self.entries):
                st = self.optimizers[oi].state.get(p)
                if st and "exp_avg" in st:
                    self.m_views[k].copy_(st["exp_avg"])
                    self.v_views[k].copy_(st["exp_avg_sq"])
                    self.steps[oi] = max(self.steps[oi], int(float(st["step"])))
        self.rebind()
        self._publish_state()

    # ---------------------------------------------------------------------------------------------- storage
    def _aliases(self, p, view):
        return p.data_ptr() == view.data_ptr() and p.device == view.device and p.dtype == view.dtype and p.is_contiguous()

    @torch.no_grad()
    def rebind(self):
        """Make every parameter a view of its flat buffer again. Cheap when nothing happened; needed after a caller replaced
        the storage of its modules (`module.to(...)`, the driver's `with to_cpu(...)` around a checkpoint)."""
        for k, (_, _, p, _, _) in enumerate(self.entries):
            if not self._aliases(p, self.p_views[k]):
                self.p_views[k].copy_(p.data)
                p.data = self.p_views[k]
            if self.shadowed:
                c = self.compute[k]
                if not self._aliases(c, self.s_views[k]):
                    self.s_views[k].copy_(p.data)
                    c.data = self.s_views[k]

    def _publish_state(self):
        for k, (oi, gi, p, _, _) in enumerate(self.entries):
            self.step_tensors[k].fill_(float(self.steps[oi]))
            self.optimizers[oi].state[p] = {"step": self.step_tensors[k], "exp_avg": self.m_views[k], "exp_avg_sq": self.v_views[k]}

    # ---------------------------------------------------------------------------------------------- one epoch
    def zero_grad(self):
        for c in self.compute:
            c.grad = None
        if self.shadowed:
            for _, _, p, _, _ in self.entries:
                p.grad = None

    @torch.no_grad()
    def collect_grads(self, which=None):
        """compute parameters' .grad -> the flat gradient buffer (one multi-tensor copy)."""
        dst, src = [], []
        for k, (oi, _, _, _, _) in enumerate(self.entries):
            if which is not None and oi not in which:
                continue
            g = self.compute[k].grad
            if g is None:
                raise RuntimeError("parameter %d of optimizer %d received no gradient: the fused step updates every parameter of the "
                                   "optimizers it serves" % (k, oi))
            dst.append(self.g_views[k])
            src.append(g)
        if dst:
            torch._foreach_copy_(dst, src)

    def all_reduce(self, which=None):
        """SUM over ranks of the flat gradient (the span of the optimizers in `which`): each rank already divided its losses by
        the GLOBAL counts. The buffer autograd's results were gathered into IS the collective's buffer."""
        if not D.is_on():
            return
        spans = [(b, e) for b, e, oi, _, _, _ in self.segments if which is None or oi in which]
        if not spans:
            return
        G = self.G[min(b for b, _ in spans):max(e for _, e in spans)]
        dev = D._comm_device(self.device)
        ev = None
        if self.time_collectives:           # HIP events around the collective (bench.py --gpus N: allreduce_ms_per_epoch)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        D._note()
        if dev == G.device:
            torch.distributed.all_reduce(G)
        else:
            tmp = G.to(dev)
            torch.distributed.all_reduce(tmp)
            G.copy_(tmp)
        if ev is not None:
            ev[1].record()
            self.collective_events.append(ev)

    def collective_ms(self):
        """[ms] of every timed gradient all-reduce since the last call (synchronises)."""
        if not self.collective_events:
            return []
        torch.cuda.synchronize(self.device)
        out = [a.elapsed_time(b) for a, b in self.collective_events]
        self.collective_events = []
        return out

    def step(self, which=None):
        """Clip + Adam for the optimizers in `which` (indices; None = all), hyper-parameters read from their param_groups now."""
        which = set(range(len(self.optimizers))) if which is None else set(which)
        for oi in which:
            self.steps[oi] += 1
        segs = (L.AdamSegment * L.ADAM_MAX_SEGMENTS)()
        n = 0
        for b, e, oi, gi, cg, mn in self.segments:
            if oi not in which:
                continue
            g = self.optimizers[oi].param_groups[gi]
            t = self.steps[oi]
            b1, b2 = g["betas"]
            s = segs[n]
            s.begin, s.end = b, e
            s.lr, s.beta1, s.beta2, s.eps, s.weight_decay = float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"])
            s.bias1, s.bias2 = 1.0 - float(b1) ** t, 1.0 - float(b2) ** t
            s.max_norm, s.clip_group = float(mn), int(cg)
            n += 1
        st = L.current_stream(self.device.index)
        ptr = lambda t: t.data_ptr() if t is not None else None
        if self.mdt == torch.float32:
            rc = self.lib.egp_adam_step_f32(n, segs, ptr(self.G), ptr(self.P), ptr(self.M), ptr(self.V), ptr(self.ws), ptr(self.norms), st)
        elif self.cdt == torch.float32:
            rc = self.lib.egp_adam_step_f64(n, segs, ptr(self.G), ptr(self.P), ptr(self.M), ptr(self.V), ptr(self.S), ptr(self.ws), ptr(self.norms), st)
        else:
            rc = self.lib.egp_adam_step_f64g(n, segs, ptr(self.G), ptr(self.P), ptr(self.M), ptr(self.V), ptr(self.ws), ptr(self.norms), st)
        L.check(rc, "egp_adam_step")
        for k, (oi, _, _, _, _) in enumerate(self.entries):
            if oi in which:
                self.step_tensors[k].fill_(float(self.steps[oi]))


class _NotEligible(Exception):
    pass
