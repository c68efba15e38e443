"""Multi-GPU glue: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm,
"gloo" in the CPU tests). The reference has no collective at all (agents/agent.py:94-106 uses a
multiprocessing.Queue inside one host); what the data-parallel PPO needs is small:

  * ONE fused all-reduce per PPO epoch over a single flat buffer holding the value-net and policy-net
    gradients (~0.48 M floats = 1.9 MB: latency-bound on a 153 GB/s xGMI link, so no bucketing);
  * scalar exchanges: advantage moments (n, mean, M2), sample counts, LoggerRL totals, ZFilter moments.

Everything degrades to a no-op when the process group is not initialised (single GPU).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def is_on():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_on() else 1


def rank():
    return dist.get_rank() if is_on() else 0


def _comm_device(device):
    """gloo moves CPU tensors; nccl/RCCL moves device tensors."""
    return torch.device("cpu") if dist.get_backend() == "gloo" else torch.device(device)


def global_count(n_local, device):
    if not is_on():
        return int(n_local)
    t = torch.tensor([float(n_local)], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t)
    _note()
    return int(round(t.item()))


def global_max(value, device="cpu"):
    if not is_on():
        return int(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    _note()
    return int(round(t.item()))


def chan_merge(parts):
    """Merge rows of (n, mean, M2) -- mean/M2 may be vectors -- in row order."""
    n, mean, m2 = 0.0, None, None
    for pn, pm, ps in parts:
        if pn <= 0:
            continue
        if n == 0.0:
            n, mean, m2 = float(pn), np.array(pm, float), np.array(ps, float)
            continue
        tot = n + pn
        d = np.asarray(pm, float) - mean
        m2 = m2 + np.asarray(ps, float) + d * d * (n * pn / tot)
        mean = mean + d * (pn / tot)
        n = tot
    return n, mean, m2


COLLECTIVES = {"count": 0}      # collectives issued through this module since the last reset (bench.py reports them per iteration)


def _note():
    COLLECTIVES["count"] += 1


def chan_merge_rows(rows, dim):
    """Chan merge, in rank order, of the rows of a device tensor (W, 1 + 2 dim) = [n | mean[dim] | M2[dim]] -> (n, mean, M2)
    tensors on the same device. Rows with n == 0 are skipped whatever their mean / M2 hold (a rank without samples may carry
    NaN there); W is the world size, so the Python loop is a handful of fused element-wise launches and nothing leaves the
    device."""
    n = rows[0, 0].clone()
    mean, m2 = rows[0, 1:1 + dim].clone(), rows[0, 1 + dim:1 + 2 * dim].clone()
    zero = torch.zeros((), dtype=rows.dtype, device=rows.device)
    empty0 = n <= 0
    mean, m2 = torch.where(empty0, zero, mean), torch.where(empty0, zero, m2)
    n = torch.where(empty0, zero, n)
    for r in range(1, rows.shape[0]):
        pn = rows[r, 0]
        use = pn > 0
        pn = torch.where(use, pn, zero)
        pm, ps = torch.where(use, rows[r, 1:1 + dim], zero), torch.where(use, rows[r, 1 + dim:1 + 2 * dim], zero)
        tot = n + pn
        safe = torch.where(tot > 0, tot, torch.ones_like(tot))
        d = pm - mean
        first = (n <= 0) & use                     # nothing merged so far: take the row as it is (exact, like chan_merge)
        m2_new = m2 + ps + d * d * (n * pn / safe)
        mean_new = mean + d * (pn / safe)
        m2 = torch.where(first, ps, torch.where(use, m2_new, m2))
        mean = torch.where(first, pm, torch.where(use, mean_new, mean))
        n = tot
    return n, mean, m2


def merge_moments_and_counts(stats, counts=None):
    """ONE float64 collective for the scalars of an update: `stats` = device float64[3] {n, mean, M2} of this rank's raw
    advantages, `counts` = this rank's sample counts (any number of them, or None). Returns (global {n, mean, M2} on the
    device, [global counts] or None). Every rank contributes the row {n, mean, M2, counts...} to one all-gather; the moments
    are Chan-merged in rank order on the device (no cancellation when |mean| >> std, ranks with n == 0 skipped even if their
    mean is NaN), the counts summed; the moments never visit the host, the counts come back in one transfer."""
    if not is_on():
        return stats, (None if counts is None else [int(c) for c in counts])
    dev = _comm_device(stats.device)
    head = stats[:3]
    if counts is not None:
        head = torch.cat((head, torch.tensor([float(c) for c in counts], dtype=torch.float64, device=stats.device)))
    mine = head.to(dev).contiguous()
    rows = torch.empty(world_size(), mine.shape[0], dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(rows, mine) if dist.get_backend() != "gloo" else dist.all_gather(list(rows.unbind(0)), mine)
    _note()
    rows = rows.to(stats.device)
    n, mean, m2 = chan_merge_rows(rows[:, :3], 1)
    out = torch.stack((n, mean[0], m2[0]))
    return out, (None if counts is None else [int(round(c)) for c in rows[:, 3:].sum(0).tolist()])


def merge_moments(stats):
    """stats: device float64[3] = {n, mean, M2} of this rank's raw advantages -> the global moments (on the device)."""
    return merge_moments_and_counts(stats, None)[0]


class FlatGradSync:
    """Gradients of a parameter list viewed through one flat buffer; one SUM all-reduce per call.

    Each rank has already divided its loss by the GLOBAL sample count, so the sum of the per-rank
    gradients is the gradient of the global-batch mean.
    """

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters to synchronise")
        p0 = self.params[0]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=p0.dtype, device=p0.device)
        self.views, pos = [], 0
        for p in self.params:
            self.views.append(self.flat[pos:pos + p.numel()].view_as(p))
            pos += p.numel()

    def attach(self):
        """Zero the flat buffer and make its views the parameters' .grad: autograd then accumulates straight into the buffer
        the collective runs on (what `zero_grad` is to the single-process update)."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce(self):
        """One SUM all-reduce over the flat buffer. Gradients that already live in it (`attach`) are not copied; any other
        .grad is copied in first and the parameter is re-pointed at the buffer."""
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()
                p.grad = v
            elif g.data_ptr() != v.data_ptr():
                v.copy_(g)
                p.grad = v
        if is_on():
            _note()
            dev = _comm_device(self.flat.device)
            if dev == self.flat.device:
                dist.all_reduce(self.flat)
            else:
                tmp = self.flat.to(dev)
                dist.all_reduce(tmp)
                self.flat.copy_(tmp)


def _logger_row(log):
    ci = np.asarray(log.total_c_info, float).ravel()
    return np.concatenate([[log.num_steps, log.num_episodes, log.total_reward, log.min_episode_reward,
                            log.max_episode_reward, log.total_c_reward, log.min_c_reward, log.max_c_reward], ci])


def _merge_logger_rows(rows):
    """LoggerRL.merge (core/logger_rl.py:44-59) over the rows of a (W, 8 + n_cinfo) tensor: sums of the totals, min / max of
    the extremes -- one row out, computed where the rows are. min_episode_reward (column 3) is the MAX of the per-worker
    minima: the reference's merge does that (core/logger_rl.py:52) and rl_core.LoggerRL.merge keeps it, so a multi-rank log
    equals the reference's merge of the same per-rank loggers."""
    sums, mins, maxs = rows.sum(0), rows.min(0).values, rows.max(0).values
    out = sums.clone()
    out[3], out[6] = maxs[3], mins[6]
    out[4], out[7] = maxs[4], maxs[7]
    return out


def merge_sampling_pass(log, running_state, base, device):
    """Everything the ranks exchange after a sampling pass in ONE collective: each rank contributes
    [LoggerRL totals (8 + n_cinfo) | what its rollout added to the observation filter (n, mean[dim], M2[dim])] to one float64
    all-gather; the logger rows are merged (core/logger_rl.py:44-59) and the filter deltas Chan-merged onto `base` in rank
    order ON the communication device, and one transfer brings the two results to the host objects that hold them
    (`log` fields are Python floats, the ZFilter's statistics NumPy arrays). Returns the merged logger; `running_state`
    (may be None) is updated in place: every rank ends up with base (+) delta_0 (+) delta_1 ..."""
    from .rl_core import LoggerRL
    if not is_on():
        return log
    lrow = _logger_row(log)
    n_log = lrow.shape[0]
    dim = 0
    parts = [lrow]
    if running_state is not None:
        rs = running_state.rs
        dim = int(np.prod(rs.shape))
        dn, dm, ds = subtract_moments((rs._n, rs._M.ravel(), rs._S.ravel()), base)
        parts += [[dn], dm, ds]
    dev = _comm_device(device)
    mine = torch.as_tensor(np.concatenate(parts), dtype=torch.float64).to(dev)
    rows = torch.empty(world_size(), mine.shape[0], dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(rows, mine) if dist.get_backend() != "gloo" else dist.all_gather(list(rows.unbind(0)), mine)
    _note()
    out = [_merge_logger_rows(rows[:, :n_log])]
    if running_state is not None:
        b = torch.as_tensor(np.concatenate([[float(base[0])], np.asarray(base[1], float).ravel(), np.asarray(base[2], float).ravel()]),
                            dtype=torch.float64).to(dev)
        n, mean, m2 = chan_merge_rows(torch.cat((b.unsqueeze(0), rows[:, n_log:]), 0), dim)
        out += [n.view(1), mean, m2]
    res = torch.cat(out).cpu().numpy()                     # the one device -> host transfer of the pass
    if running_state is not None:
        rs = running_state.rs
        rs._n = int(round(res[n_log]))
        rs._M = res[n_log + 1:n_log + 1 + dim].reshape(rs.shape).copy()
        rs._S = res[n_log + 1 + dim:n_log + 1 + 2 * dim].reshape(rs.shape).copy()
    return LoggerRL.from_totals(*res[:8].tolist(), res[8:n_log])


def merge_loggers(log, device):
    """LoggerRL.merge across ranks (core/logger_rl.py:44-59); one collective (merge_sampling_pass without a filter)."""
    return merge_sampling_pass(log, None, None, device)


def subtract_moments(total, base):
    """Inverse Chan merge: the (n, mean, M2) of the samples added on top of ``base`` to reach ``total``."""
    nt, mt, st = total
    nb, mb, sb = base
    nd = nt - nb
    if nd <= 0:
        return 0.0, np.zeros_like(np.asarray(mt, float)), np.zeros_like(np.asarray(st, float))
    if nb <= 0:
        return float(nt), np.array(mt, float), np.array(st, float)
    md = (nt * np.asarray(mt, float) - nb * np.asarray(mb, float)) / nd
    d = md - np.asarray(mb, float)
    sd = np.asarray(st, float) - np.asarray(sb, float) - d * d * (nb * nd / nt)
    return float(nd), md, np.maximum(sd, 0.0)


def merge_running_state(running_state, base, device):
    """Every rank ends up with base (+) delta_0 (+) delta_1 ... where delta_r is what rank r's rollout added to the
    observation filter during this sampling pass (merge_sampling_pass with a throw-away logger row)."""
    if not is_on():
        return
    from .rl_core import LoggerRL
    merge_sampling_pass(LoggerRL.from_totals(1, 1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, np.zeros(1)), running_state, base, device)


def init_from_env(device_index=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); returns (rank, world, local_rank)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, int(os.environ.get("LOCAL_RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not is_on():
        backend = os.environ.get("EGP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            n_dev = max(1, torch.cuda.device_count())
            torch.cuda.set_device((local if device_index is None else device_index) % n_dev)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
    count_ranks_on_host()
    return dist.get_rank(), dist.get_world_size(), local


def count_ranks_on_host():
    """How many ranks of the group run on this host (they share its cores: physics.default_threads): one all-gather of the
    host names at a point EVERY rank passes -- group set-up -- cached for `physics.ranks_on_host()`, which only reads it.
    Errors of the collective propagate."""
    import socket
    from . import physics
    if not is_on():
        return 1
    names = [None] * dist.get_world_size()
    dist.all_gather_object(names, socket.gethostname())
    physics._RANKS_ON_HOST = max(1, sum(1 for n in names if n == socket.gethostname()))
    return physics._RANKS_ON_HOST
