"""world_size-2 `gloo` runs on CPU: the data-parallel PPO update (one fused flat-gradient all-reduce per epoch,
global advantage moments, global sample counts) must reproduce the single-process full-batch reference run,
and the small scalar exchanges (logger totals, observation-filter moments) must merge exactly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_default_dtype(torch.float64)
    torch.set_num_threads(1)
    from conftest import load_golden
    from update_fixture import build_agent
    from egopose_amd import dist as D
    from egopose_amd.rl_core import TrajBatchEgo, LoggerRL
    from egopose_amd.zfilter import ZFilter
    from oracle.gae import estimate_advantages as oracle_gae

    g = load_golden("ppo_update.npz")
    agent, mods = build_agent(g)
    # shard the flat batch at an episode boundary: episodes [10,4,7] -> rank 0, [10,2,10,5] -> rank 1
    cut = 21
    sl = slice(0, cut) if rank == 0 else slice(cut, None)
    cols = {k: torch.as_tensor(g[k][sl]) for k in ("states", "actions", "masks", "rewards", "exps", "v_metas")}
    cols["next_states"] = cols["states"]
    batch = TrajBatchEgo.from_device(**cols)

    def adv_fn(rewards, masks, values):
        _, ret, raw = oracle_gae(rewards.numpy(), masks.numpy(), values.numpy(), agent.gamma, agent.tau)
        raw = raw.ravel()
        st = torch.tensor([raw.size, raw.mean(), ((raw - raw.mean()) ** 2).sum()], dtype=torch.float64)
        st = D.merge_moments(st)                                   # global {n, mean, M2}
        a = (raw - st[1].item()) / np.sqrt(st[2].item() / (st[0].item() - 1.0))
        return torch.as_tensor(a).unsqueeze(1), torch.as_tensor(ret)
    agent._advantages = adv_fn
    agent.update_params(batch)
    final = {"%s__%s" % (n, k): v.numpy() for n, m in mods.items() for k, v in m.state_dict().items()}

    # scalar exchanges
    lg = LoggerRL.from_totals(10 + rank, 2 + rank, 10.0 + rank, 3 + rank, 7 + rank, 4.5 * (rank + 1), 0.1 * (rank + 1), 0.9 - 0.1 * rank,
                              np.arange(5.0) * (rank + 1))
    rng = np.random.RandomState(7)
    X = rng.normal(size=(60, 5)) * 2 + 1
    zf = ZFilter((5,), clip=5)
    for x in X[:10]:
        zf(x)                                                      # common history
    base = (float(zf.rs._n), zf.rs._M.copy(), zf.rs._S.copy())
    mine = X[10:35] if rank == 0 else X[35:]
    zf.rs.merge(len(mine), mine.mean(0), ((mine - mine.mean(0)) ** 2).sum(0))
    before = D.COLLECTIVES["count"]
    merged = D.merge_sampling_pass(lg, zf, base, "cpu")           # logger totals + filter deltas: ONE collective
    n_coll = D.COLLECTIVES["count"] - before
    # the two single-purpose wrappers give the same numbers
    merged_b = D.merge_loggers(lg, "cpu")
    assert merged_b.num_steps == merged.num_steps and merged_b.min_c_reward == merged.min_c_reward
    # advantage moments: |mean| >> std and a rank without samples whose mean is NaN (raw sums cancel / poison)
    big = torch.tensor([4.0, 1.0e9 + rank, 2.0], dtype=torch.float64) if rank == 0 else torch.tensor([0.0, float("nan"), float("nan")], dtype=torch.float64)
    gm, cnt = D.merge_moments_and_counts(big, (3 + rank, 1))
    assert cnt == [7, 2]
    np.testing.assert_array_equal(gm.numpy(), [4.0, 1.0e9, 2.0])  # the empty rank changes nothing, nothing cancels
    two = torch.tensor([3.0, 1.0e8 + 1.0, 2.0], dtype=torch.float64) if rank == 0 else torch.tensor([2.0, 1.0e8 + 3.5, 0.5], dtype=torch.float64)
    gm2 = D.merge_moments(two)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), n_steps=merged.num_steps, avg_c=merged.avg_c_reward,
             min_c=merged.min_c_reward, max_ep=merged.max_episode_reward, min_ep=merged.min_episode_reward, avg_ci=merged.avg_c_info,
             zf_n=zf.rs.n, zf_mean=zf.rs.mean, zf_std=zf.rs.std, count=D.global_count(3 + rank, "cpu"),
             gmax=D.global_max(5 + 2 * rank), n_coll=n_coll, gm2=gm2.numpy(), **final)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ppo_update_equals_single_process_reference(tmp_path):
    from conftest import load_golden
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = load_golden("ppo_update.npz")
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in g.files:
        if not k.startswith("final_"):
            continue
        key = k[len("final_"):]
        np.testing.assert_allclose(r0[key], g[k], rtol=1e-9, atol=1e-10, err_msg=key)     # == reference full batch
        np.testing.assert_array_equal(r0[key], r1[key])                                      # ranks stay bit-identical
    assert int(r0["n_steps"]) == 21 and int(r0["count"]) == 7 and int(r0["gmax"]) == 7
    np.testing.assert_allclose(r0["avg_c"], (4.5 + 9.0) / 21)
    np.testing.assert_allclose(r0["min_c"], 0.1)
    # per-rank minima 3 and 4, maxima 7 and 8: the reference's merge takes max() of the MINIMA too (core/logger_rl.py:52)
    assert float(r0["min_ep"]) == 4.0 and float(r0["max_ep"]) == 8.0 and float(r1["min_ep"]) == 4.0
    np.testing.assert_allclose(r0["avg_ci"], np.arange(5.0) * 3 / 21)
    assert int(r0["n_coll"]) == 1 and int(r1["n_coll"]) == 1                  # one collective per sampling pass
    # Chan merge of {3, 1e8+1, 2} and {2, 1e8+3.5, 0.5}: d = 2.5, M2 = 2.5 + 6.25 * 6/5 = 10 (raw sums would lose it at 1e16)
    np.testing.assert_allclose(r0["gm2"], [5.0, 1.0e8 + 2.0, 10.0], rtol=1e-14)
    np.testing.assert_array_equal(r0["gm2"], r1["gm2"])
    # observation filter: merged moments == pushing all 60 rows sequentially
    from egopose_amd.zfilter import ZFilter
    rng = np.random.RandomState(7)
    X = rng.normal(size=(60, 5)) * 2 + 1
    ref = ZFilter((5,), clip=5)
    for x in X:
        ref(x)
    for r in (r0, r1):
        assert int(r["zf_n"]) == 60
        np.testing.assert_allclose(r["zf_mean"], ref.rs.mean, rtol=1e-12)
        np.testing.assert_allclose(r["zf_std"], ref.rs.std, rtol=1e-11)
