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
    return int(round(t.item()))


def global_max(value, device="cpu"):
    if not is_on():
        return int(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
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


def merge_moments_and_counts(stats, counts=None):
    """ONE float64 all-reduce for the scalars of an update: `stats` = device float64[3] {n, mean, M2} of this rank's raw
    advantages, `counts` = this rank's sample counts (any number of them, or None). Returns (global {n, mean, M2} on the
    device, [global counts] or None). Every rank sends {n, n mean, M2 + n mean^2, counts...}; from the sums
    mean = S1 / N and M2 = S2 - N mean^2 (Chan's merge written as sums: exact for one rank, and the subtraction loses
    nothing that matters at float64 -- advantages have |mean| << std). The result stays on the device (no host copy of the
    moments); only the counts come back to the host, in the same transfer."""
    if not is_on():
        return stats, (None if counts is None else [int(c) for c in counts])
    dev = _comm_device(stats.device)
    n, mean, m2 = stats[0], stats[1], stats[2]
    head = torch.stack((n, n * mean, m2 + n * mean * mean))
    if counts is not None:
        head = torch.cat((head, torch.tensor([float(c) for c in counts], dtype=torch.float64, device=stats.device)))
    vec = head.to(dev)
    dist.all_reduce(vec)
    vec = vec.to(stats.device)
    gm = vec[1] / vec[0]
    out = torch.stack((vec[0], gm, torch.clamp(vec[2] - vec[0] * gm * gm, min=0.0)))
    return out, (None if counts is None else [int(round(c)) for c in vec[3:].tolist()])


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
            dev = _comm_device(self.flat.device)
            if dev == self.flat.device:
                dist.all_reduce(self.flat)
            else:
                tmp = self.flat.to(dev)
                dist.all_reduce(tmp)
                self.flat.copy_(tmp)


def merge_loggers(log, device):
    """LoggerRL.merge across ranks (core/logger_rl.py:44-59)."""
    from .rl_core import LoggerRL
    if not is_on():
        return log
    ci = np.asarray(log.total_c_info, float).ravel()
    row = np.concatenate([[log.num_steps, log.num_episodes, log.total_reward, log.min_episode_reward,
                           log.max_episode_reward, log.total_c_reward, log.min_c_reward, log.max_c_reward], ci])
    dev = _comm_device(device)
    mine = torch.as_tensor(row, dtype=torch.float64, device=dev)
    rows = [torch.empty_like(mine) for _ in range(world_size())]
    dist.all_gather(rows, mine)
    parts = [LoggerRL.from_totals(*r[:8].tolist(), r[8:].numpy()) for r in (x.cpu() for x in rows)]
    return LoggerRL.merge(parts)


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
    """Every rank ends up with base (+) delta_0 (+) delta_1 ... where delta_r is what rank r's rollout
    added to the observation filter during this sampling pass."""
    if not is_on():
        return
    rs = running_state.rs
    dim = int(np.prod(rs.shape))
    dn, dm, ds = subtract_moments((rs._n, rs._M.ravel(), rs._S.ravel()), base)
    dev = _comm_device(device)
    mine = torch.as_tensor(np.concatenate([[dn], dm, ds]), dtype=torch.float64, device=dev)
    rows = [torch.empty_like(mine) for _ in range(world_size())]
    dist.all_gather(rows, mine)
    parts = [base] + [(float(r[0]), r[1:1 + dim].cpu().numpy(), r[1 + dim:].cpu().numpy()) for r in rows]
    n, mean, m2 = chan_merge(parts)
    rs._n = int(round(n))
    rs._M = np.asarray(mean, float).reshape(rs.shape).copy()
    rs._S = np.asarray(m2, float).reshape(rs.shape).copy()


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
    return dist.get_rank(), dist.get_world_size(), local
