"""ORACLE (test infrastructure): the reference's CPU trajectory sampler, restated.

One env per worker, batch-1 float64 torch policy on CPU, per-sample observation filter,
per-step custom reward -- the structure of ``Agent.sample`` / ``sample_worker``
(/root/reference/agents/agent.py:29-111) with the ``AgentEgo`` hooks
(/root/reference/ego_pose/core/agent_ego.py:18-32), ``Memory``/``TrajBatch(Ego)`` stacking
(utils/memory.py:4-23, core/trajbatch.py:4-16, ego_pose/core/trajbatch_ego.py:5-9) and
``LoggerRL`` bookkeeping (core/logger_rl.py:4-59).

Used (a) by tests to pin the sampler semantics against tests/golden/sampler_toy.npz and
(b) by bench.py's ``cpu_baseline`` leg ("port"): it is what the GPU rollout is timed beside,
never what is shipped.
"""
import copy
import math
import multiprocessing
import time

import numpy as np
import torch


class LogOracle:
    FIELDS = ("num_steps", "num_episodes", "total_reward", "min_episode_reward", "max_episode_reward",
              "total_c_reward", "min_c_reward", "max_c_reward", "avg_episode_reward", "avg_c_reward")

    def __init__(self):
        self.num_steps = 0
        self.num_episodes = 0
        self.total_reward = 0.0
        self.min_episode_reward = math.inf
        self.max_episode_reward = -math.inf
        self.total_c_reward = 0.0
        self.min_c_reward = math.inf
        self.max_c_reward = -math.inf
        self.total_c_info = 0.0
        self.avg_episode_reward = 0.0
        self.avg_c_reward = 0.0
        self.avg_c_info = 0.0
        self.sample_time = 0.0
        self._ep = 0.0

    def step(self, env_reward, c_reward, c_info):
        self._ep += env_reward
        self.total_c_reward += c_reward
        self.total_c_info = self.total_c_info + c_info
        self.min_c_reward = min(self.min_c_reward, c_reward)
        self.max_c_reward = max(self.max_c_reward, c_reward)
        self.num_steps += 1

    def end_episode(self):
        self.num_episodes += 1
        self.total_reward += self._ep
        self.min_episode_reward = min(self.min_episode_reward, self._ep)
        self.max_episode_reward = max(self.max_episode_reward, self._ep)
        self._ep = 0.0

    def finish(self):
        self.avg_episode_reward = self.total_reward / self.num_episodes
        self.avg_c_reward = self.total_c_reward / self.num_steps
        self.avg_c_info = self.total_c_info / self.num_steps

    @staticmethod
    def merge(logs):
        """core/logger_rl.py:44-59 -- note min_episode_reward is merged with max() there."""
        m = LogOracle()
        m.total_reward = sum(x.total_reward for x in logs)
        m.num_episodes = sum(x.num_episodes for x in logs)
        m.num_steps = sum(x.num_steps for x in logs)
        m.avg_episode_reward = m.total_reward / m.num_episodes
        m.max_episode_reward = max(x.max_episode_reward for x in logs)
        m.min_episode_reward = max(x.min_episode_reward for x in logs)
        m.total_c_reward = sum(x.total_c_reward for x in logs)
        m.avg_c_reward = m.total_c_reward / m.num_steps
        m.max_c_reward = max(x.max_c_reward for x in logs)
        m.min_c_reward = min(x.min_c_reward for x in logs)
        m.total_c_info = sum(x.total_c_info for x in logs)
        m.avg_c_info = m.total_c_info / m.num_steps
        return m


def stack_batch(rows_per_worker):
    """Concatenate worker memories in pid order and column-stack (TrajBatch / TrajBatchEgo)."""
    rows = [r for w in rows_per_worker for r in w]
    cols = list(zip(*rows))
    names = ["states", "actions", "masks", "next_states", "rewards", "exps", "v_metas"][:len(cols)]
    return {n: np.stack(c) for n, c in zip(names, cols)}


def _worker(pid, quota, env, select_action, running_state, custom_reward, noise_rate, mean_action_flag,
            pre_episode, v_meta_fn):
    torch.randn(pid)
    if hasattr(env, "np_random"):
        env.np_random.rand(pid)
    rows, log = [], LogOracle()
    while log.num_steps < quota:
        state = env.reset()
        if running_state is not None:
            state = running_state(state)
        if pre_episode is not None:
            pre_episode(env)
        for t in range(10000):
            use_mean = mean_action_flag or np.random.binomial(1, 1 - noise_rate)
            action = select_action(state, t, bool(use_mean)).astype(np.float64)
            nxt, env_r, done, info = env.step(action)
            if running_state is not None:
                nxt = running_state(nxt)
            if custom_reward is not None:
                c_r, c_info = custom_reward(env, state, action, info)
                reward = c_r
            else:
                c_r, c_info, reward = 0.0, np.array([0.0]), env_r
            log.step(env_r, c_r, c_info)
            row = [state, action, 0 if done else 1, nxt, reward, 1 - use_mean]
            if v_meta_fn is not None:
                row.append(v_meta_fn(env))
            rows.append(row)
            if done:
                break
            state = nxt
        log.end_episode()
    log.finish()
    return rows, log


def sample(min_batch_size, num_threads, env, select_action, running_state=None, custom_reward=None,
           noise_rate=1.0, mean_action=False, pre_episode=None, v_meta_fn=None, use_fork=True):
    """Returns (batch dict, merged LogOracle). ``select_action(state, t, use_mean) -> np.ndarray``."""
    t0 = time.time()
    quota = int(math.floor(min_batch_size / num_threads))
    args = (env, select_action, running_state, custom_reward, noise_rate, mean_action, pre_episode, v_meta_fn)
    results = [None] * num_threads
    with torch.no_grad():
        if use_fork and num_threads > 1:
            ctx = multiprocessing.get_context("fork")
            queue = ctx.Queue()

            def child(pid):
                queue.put((pid,) + _worker(pid, quota, *args))
            procs = [ctx.Process(target=child, args=(pid,)) for pid in range(1, num_threads)]
            for p in procs:
                p.start()
            results[0] = _worker(0, quota, *args)
            for _ in procs:
                pid, rows, log = queue.get()
                results[pid] = (rows, log)
            for p in procs:
                p.join()
        else:
            # fork emulation: every worker starts from the state the parent had at "fork" time
            snap_t, snap_n = torch.get_rng_state(), np.random.get_state()
            snap_args = copy.deepcopy(args) if num_threads > 1 else None
            for pid in range(num_threads - 1, 0, -1):
                torch.set_rng_state(snap_t)
                np.random.set_state(snap_n)
                results[pid] = _worker(pid, quota, *copy.deepcopy(snap_args))
            torch.set_rng_state(snap_t)
            np.random.set_state(snap_n)
            results[0] = _worker(0, quota, *args)
    batch = stack_batch([r[0] for r in results])
    log = LogOracle.merge([r[1] for r in results])
    log.sample_time = time.time() - t0
    return batch, log
