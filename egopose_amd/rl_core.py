"""Trajectory containers and sampling statistics of the PPO loop.

Drop-in surface of /root/reference/utils/memory.py:4-23 (``Memory``), core/trajbatch.py:4-16 (``TrajBatch``),
ego_pose/core/trajbatch_ego.py:5-9 (``TrajBatchEgo``) and core/logger_rl.py:4-59 (``LoggerRL``).

The lockstep rollout never builds per-step Python tuples: it hands the batch over as episode-major
arrays that already live in HBM (``TrajBatchEgo.from_device``). The numpy attributes the reference
exposes (``states, actions, masks, next_states, rewards, exps, v_metas``; float64 / ints) are
materialised lazily, and ``update_params`` reads the device copies directly.
"""
from __future__ import annotations

import math
import random

import numpy as np

_COLUMNS = ("states", "actions", "masks", "next_states", "rewards", "exps")


class Memory:
    """Append-only list of per-step records (kept for callers that sample env by env)."""

    def __init__(self):
        self.memory = []

    def push(self, *fields):
        self.memory.append(list(fields))

    def append(self, other):
        self.memory.extend(other.memory)

    def sample(self, batch_size=None):
        return self.memory if batch_size is None else random.sample(self.memory, batch_size)

    def __len__(self):
        return len(self.memory)


class TrajBatch:
    columns = _COLUMNS

    def __init__(self, memory_list=None):
        self._dev = {}
        self._np = {}
        if memory_list:
            rows = [r for mem in memory_list for r in mem.sample()]
            for i, name in enumerate(self.columns):
                self._np[name] = np.stack([r[i] for r in rows])

    @classmethod
    def from_device(cls, **tensors):
        """Episode-major device tensors (torch, float64/int64) -> batch without a host copy."""
        b = cls()
        missing = [c for c in cls.columns if c not in tensors]
        if missing:
            raise ValueError("missing batch columns: %s" % missing)
        b._dev = dict(tensors)
        return b

    def device_column(self, name):
        return self._dev.get(name)

    def __getattr__(self, name):
        if name.startswith("_") or name not in type(self).columns:
            raise AttributeError(name)
        if name not in self._np:
            if name not in self._dev:
                raise AttributeError(name)
            self._np[name] = self._dev[name].detach().cpu().numpy()
        return self._np[name]

    def __len__(self):
        c = self.columns[0]
        return int(self._dev[c].shape[0]) if c in self._dev else int(self._np[c].shape[0])


class TrajBatchEgo(TrajBatch):
    columns = _COLUMNS + ("v_metas",)


class LoggerRL:
    """Sampling statistics of one worker (one env slot, or a whole lockstep pass via ``from_totals``) and their ``merge``.

    The attribute schema is the reference's (core/logger_rl.py:6-20): the unmodified driver reads ``sample_time``,
    ``avg_c_reward``, ``avg_c_info``, ``min_c_reward`` / ``max_c_reward`` and ``avg_episode_reward`` off the merged object
    (ego_pose/ego_mimic.py:120-131). The schema lives in three tables here -- additive counters, running extrema, derived
    averages -- and every method is a loop over them."""

    # additive over steps / episodes / workers
    _SUMS = ("total_reward", "num_episodes", "num_steps", "total_c_reward", "total_c_info")
    # running extrema: (attribute, reducer inside a worker, reducer across workers). The reference folds the workers'
    # minimum episode rewards with max() (core/logger_rl.py:52); kept, the driver's printout depends on it.
    _EXTREMA = (("min_episode_reward", min, max), ("max_episode_reward", max, max),
                ("min_c_reward", min, min), ("max_c_reward", max, max))
    # derived at the end of a sampling pass: attribute = numerator / denominator
    _AVERAGES = (("avg_episode_reward", "total_reward", "num_episodes"), ("avg_c_reward", "total_c_reward", "num_steps"),
                 ("avg_c_info", "total_c_info", "num_steps"))

    def __init__(self):
        for name in self._SUMS:
            setattr(self, name, 0)
        for name, inner, _ in self._EXTREMA:
            setattr(self, name, math.inf if inner is min else -math.inf)
        for name, _, _ in self._AVERAGES:
            setattr(self, name, 0)
        self.episode_reward = 0          # the running episode's return
        self.sample_time = 0             # filled in by the sampler

    def _fold(self, name, value):
        for attr, inner, _ in self._EXTREMA:
            if attr == name:
                setattr(self, attr, inner(getattr(self, attr), value))

    def start_episode(self, env):
        self.episode_reward = 0

    def step(self, env, reward, c_reward, c_info):
        self.num_steps += 1
        self.episode_reward += reward
        self.total_c_reward += c_reward
        self.total_c_info += c_info
        self._fold("min_c_reward", c_reward)
        self._fold("max_c_reward", c_reward)

    def end_episode(self, env):
        ret = self.episode_reward
        self.num_episodes += 1
        self.total_reward += ret
        self._fold("min_episode_reward", ret)
        self._fold("max_episode_reward", ret)

    def end_sampling(self):
        self._averages()

    def _averages(self):
        for name, num, den in self._AVERAGES:
            setattr(self, name, getattr(self, num) / getattr(self, den))

    @classmethod
    def from_totals(cls, num_steps, num_episodes, total_reward, min_ep, max_ep, total_c_reward, min_c, max_c, total_c_info):
        """The statistics of a whole lockstep pass, reduced on the device (rollout.LockstepRollout)."""
        lg = cls()
        lg.num_steps, lg.num_episodes, lg.total_reward = int(num_steps), int(num_episodes), float(total_reward)
        lg.min_episode_reward, lg.max_episode_reward = float(min_ep), float(max_ep)
        lg.total_c_reward, lg.min_c_reward, lg.max_c_reward = float(total_c_reward), float(min_c), float(max_c)
        lg.total_c_info = np.asarray(total_c_info, dtype=float)
        lg._averages()
        return lg

    @classmethod
    def merge(cls, logger_list):
        out = cls()
        for name in cls._SUMS:
            setattr(out, name, sum(getattr(x, name) for x in logger_list))
        for name, _, across in cls._EXTREMA:
            setattr(out, name, across(getattr(x, name) for x in logger_list))
        out._averages()
        return out
