"""The unmodified reference driver on the GPU: ego_pose/ego_mimic.py:31-32,52-69,83-90 builds float64 nets, plain
torch optimizers over them and `AgentEgo(dtype=torch.float64, device=cuda, ...)` with float64 as the process-wide default
dtype. (The driver file itself is executed in the build container, tests/test_host_logic.py; /root/reference does not
exist on the GPU box.) That set-up must run on the fast path: float32 shadow nets through the fused policy kernel, the
fast tick and the persistent HIP LSTM launches, float64 masters / state_dict / TrajBatch."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workspace(tmp_path_factory):
    from egopose_amd.bench_support import write_synthetic_dataset
    root = str(tmp_path_factory.mktemp("egp_dropin"))
    write_synthetic_dataset(root, "subject_03", n_takes=3, n_frames=300, seed=4)
    return root


def test_float64_driver_setup_runs_on_the_float32_hip_kernels(workspace, monkeypatch):
    from egopose_amd import lstm
    from egopose_amd.config import Config
    from egopose_amd.train import Trainer
    group_calls = []
    inner_group = lstm.lstm_group
    monkeypatch.setattr(lstm, "lstm_group", lambda *a, **k: (group_calls.append(1), inner_group(*a, **k))[1])
    os.chdir(workspace)
    torch.set_default_dtype(torch.float64)                     # ego_mimic.py:31-32
    try:
        cfg = Config("subject_03", create_dirs=False)
        cfg.env_episode_len = 14
        cfg.num_optim_epoch = 2
        tr = Trainer(cfg, torch.device("cuda", 0), torch.float64, num_envs=48, num_threads=4, num_groups=2, plain_optim=True)
        agent = tr.agent
        assert agent.dtype == torch.float64 and agent.cdtype == torch.float32 and agent.shadow is not None
        masters = [tr.policy_net, tr.policy_vs_net, tr.value_net, tr.value_vs_net]
        assert all(p.dtype == torch.float64 for m in masters for p in m.parameters())
        assert not any(getattr(g, "fused", None) for g in tr.optimizer_policy.param_groups)      # the driver's plain Adam
        before = [p.detach().clone() for m in masters for p in m.parameters()]
        log, t_s, t_u, n = tr.iteration(0, 48 * 20)
        ro = agent._get_rollout()
        assert ro._fused is not None, "the fused HIP policy step must serve the float64 driver set-up"
        assert ro.policy_net is agent.cn.policy_net and next(ro.policy_net.parameters()).dtype == torch.float32
        assert ro.v_out.dtype == torch.float32
        assert len(group_calls) >= 1 + 2, "persistent HIP LSTM launches: context pool, 2 epochs (the first doubles as the value pass)"
        assert n >= 48 * 20 and np.isfinite(log.avg_c_reward)
        batch, _ = agent.sample(48 * 20)
        for k in ("states", "actions", "next_states", "rewards"):
            assert getattr(batch, k).dtype == np.float64, k
        after = [p for m in masters for p in m.parameters()]
        assert all(p.dtype == torch.float64 for p in after)
        assert all(not torch.equal(a, b) for a, b in zip(before, after) if b.requires_grad)
        for m, s in agent.shadow.pairs:
            assert torch.equal(s, m.float())
        # the driver's per-iteration writes reach the kernels: action_log_std.fill_ on the MASTER (ego_mimic.py:97-98)
        with torch.no_grad():
            tr.policy_net.action_log_std.fill_(-3.1)
        agent.mean_action = False
        batch2, _ = agent.sample(48 * 20)
        assert float(agent.cn.policy_net.action_log_std[0, 0]) == pytest.approx(-3.1, rel=1e-6)
        assert ro._fused.log_std[0].item() == pytest.approx(-3.1, rel=1e-6)
        # checkpoint as the driver writes it (ego_mimic.py:133-139): float64 state_dicts, reference-named running_state
        path = os.path.join(workspace, "cp_dropin.p")
        tr.save(path)
        from egopose_amd.zfilter import reference_pickle_names
        with reference_pickle_names():
            cp = pickle.load(open(path, "rb"))
        assert all(v.dtype == torch.float64 for v in cp["policy_vs_dict"].values()) and cp["running_state"].rs.n > 0
        # the masters moved to the CPU and back (`with to_cpu(...)`): the next iteration still runs and still learns
        log2, *_ = tr.iteration(1, 48 * 20)
        assert np.isfinite(log2.avg_c_reward)
        tr.close()
    finally:
        torch.set_default_dtype(torch.float32)


def test_net_dtype_override_keeps_float64_compute(workspace, monkeypatch):
    """EGP_NET_DTYPE=float64: no shadows, the float64 torch paths (parity runs)."""
    from egopose_amd.config import Config
    from egopose_amd.train import Trainer
    monkeypatch.setenv("EGP_NET_DTYPE", "float64")
    os.chdir(workspace)
    torch.set_default_dtype(torch.float64)
    try:
        cfg = Config("subject_03", create_dirs=False)
        cfg.env_episode_len = 8
        cfg.num_optim_epoch = 1
        tr = Trainer(cfg, torch.device("cuda", 0), torch.float64, num_envs=16, num_threads=2, num_groups=1, plain_optim=True)
        assert tr.agent.shadow is None and tr.agent.cdtype == torch.float64
        log, t_s, t_u, n = tr.iteration(0, 16 * 10)
        assert n >= 160 and np.isfinite(log.avg_c_reward) and tr.agent._get_rollout()._fused is None
        tr.close()
    finally:
        torch.set_default_dtype(torch.float32)
