"""Training entry point of the MI355X path: the flow of the reference's ego_pose/ego_mimic.py (:28-147) --
config -> env + experts -> nets + optimizers -> AgentEgo -> {sample, update, log, checkpoint} -- written
against this package so it also serves bench.py, smoke() and the GPU tests.

    python -m egopose_amd.train --cfg subject_03 --data <dir with config/ and datasets/> [--synthetic]

The unmodified reference driver runs on the same classes through ``egopose_amd/compat``.
"""
from __future__ import annotations

import argparse
import os
import pickle
import time

import numpy as np
import torch

from . import dist as D
from . import gemm_tuning
from .agent import AgentEgo, compute_dtype
from .config import Config, ForecastConfig
from .env import HumanoidEnv
from .logging_utils import Logger, create_logger
from .nets import MLP, PolicyGaussian, Value, VideoForecastNet, VideoStateNet
from .reward import reward_func
from .torch_utils import set_optimizer_lr, to_cpu, to_device
from .zfilter import ZFilter, dump_reference_pickle, load_reference_pickle


class Trainer:
    """Everything ego_mimic.py builds at module level, as one object."""

    def __init__(self, cfg, device, dtype=torch.float32, num_envs=1024, num_threads=None, num_groups=2, seed_offset=0,
                 plain_optim=False):
        """`dtype=torch.float64` is the reference driver's set-up (ego_pose/ego_mimic.py:31-32): float64 modules and
        state_dicts; on a GPU the agent computes with float32 shadow copies (agent.ShadowNets). `plain_optim=True`
        builds the optimizers exactly as the driver does (no fused Adam)."""
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.plain_optim = bool(plain_optim)
        if self.device.type == "cuda" and compute_dtype(dtype, self.device) == torch.float32 and not gemm_tuning.enabled():
            gemm_tuning.enable()                         # bucketed update shapes + pre-tuned rocBLAS/hipBLASLt picks
        np.random.seed(cfg.seed + seed_offset)
        torch.manual_seed(cfg.seed)                      # identical initial weights on every rank
        env = HumanoidEnv(cfg)
        env.seed(cfg.seed + seed_offset)
        env.load_experts(cfg.takes["train"], cfg.expert_feat_file, cfg.cnn_feat_file)
        self.env = env
        cnn_dim = env.cnn_feat[0].shape[-1]
        state_dim, action_dim = env.observation_space.shape[0], env.action_space.shape[0]
        self.running_state = ZFilter((state_dim,), clip=5)
        self.forecast = getattr(cfg, "task", "egomimic") == "egoforecast"
        if self.forecast:        # ego_pose/ego_forecast.py:53-59: causal video net over the past + per-step state net
            mk_vs = lambda p: VideoForecastNet(cnn_dim, state_dim, getattr(cfg, p + "_v_hdim"), cfg.fr_margin, getattr(cfg, p + "_v_net"),
                                               getattr(cfg, p + "_v_net_param"), getattr(cfg, p + "_s_hdim"), getattr(cfg, p + "_s_net"),
                                               getattr(cfg, p + "_dyn_v"))
            self.policy_vs_net, self.value_vs_net = mk_vs("policy"), mk_vs("value")
            p_in, v_in = self.policy_vs_net.out_dim, self.value_vs_net.out_dim
        else:
            mk_vs = lambda hdim, kind, param: VideoStateNet(cnn_dim, hdim, cfg.fr_margin, kind, param, cfg.causal)
            self.policy_vs_net = mk_vs(cfg.policy_v_hdim, cfg.policy_v_net, cfg.policy_v_net_param)
            self.value_vs_net = mk_vs(cfg.value_v_hdim, cfg.value_v_net, cfg.value_v_net_param)
            p_in, v_in = state_dim + cfg.policy_v_hdim, state_dim + cfg.value_v_hdim
        self.policy_net = PolicyGaussian(MLP(p_in, cfg.policy_hsize, cfg.policy_htype), action_dim,
                                         log_std=cfg.log_std, fix_std=cfg.fix_std)
        self.value_net = Value(MLP(v_in, cfg.value_hsize, cfg.value_htype))
        self.nets = dict(policy_dict=self.policy_net, policy_vs_dict=self.policy_vs_net, value_dict=self.value_net,
                         value_vs_dict=self.value_vs_net)
        for net in self.nets.values():
            net.to(dtype)
        to_device(self.device, *self.nets.values())
        policy_params = list(self.policy_net.parameters()) + list(self.policy_vs_net.parameters())
        value_params = list(self.value_net.parameters()) + list(self.value_vs_net.parameters())
        self.optimizer_policy = self._optimizer(cfg.policy_optimizer, policy_params, cfg.policy_lr, cfg.policy_momentum, cfg.policy_weightdecay)
        self.optimizer_value = self._optimizer(cfg.value_optimizer, value_params, cfg.value_lr, cfg.value_momentum, cfg.value_weightdecay)
        self.agent = AgentEgo(env=env, dtype=dtype, device=self.device, running_state=self.running_state,
                              custom_reward=reward_func[cfg.reward_id], mean_action=False, render=False,
                              num_threads=num_threads, num_envs=num_envs, num_groups=num_groups,
                              policy_net=self.policy_net, policy_vs_net=self.policy_vs_net, value_net=self.value_net,
                              value_vs_net=self.value_vs_net, optimizer_policy=self.optimizer_policy,
                              optimizer_value=self.optimizer_value, opt_num_epochs=cfg.num_optim_epoch, gamma=cfg.gamma,
                              tau=cfg.tau, clip_epsilon=cfg.clip_epsilon, policy_grad_clip=[(policy_params, 40)])

    def _optimizer(self, kind, params, lr, momentum, weight_decay):
        if kind == "Adam":
            # built exactly as the driver builds them (ego_pose/ego_mimic.py:70-77); on the GPU the agent steps them through
            # optim.FlatUpdater (clip + both Adam steps in two launches over flat buffers, same update rule)
            return torch.optim.Adam(list(params), lr=lr, weight_decay=weight_decay)
        return torch.optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay)

    def pre_iter_update(self, i_iter):
        cfg = self.cfg
        cfg.update_adaptive_params(i_iter)
        if self.forecast:
            cfg.env_init_noise = cfg.adp_init_noise           # ego_forecast.py:108
        self.agent.set_noise_rate(cfg.adp_noise_rate)
        set_optimizer_lr(self.optimizer_policy, cfg.adp_policy_lr)
        if cfg.fix_std:
            self.policy_net.action_log_std.data.fill_(cfg.adp_log_std)

    def iteration(self, i_iter, min_batch_size=None):
        """One PPO iteration; returns (LoggerRL, T_sample, T_update, env_steps of this rank)."""
        cfg = self.cfg
        self.pre_iter_update(i_iter)
        batch, log = self.agent.sample(cfg.min_batch_size if min_batch_size is None else min_batch_size)
        if getattr(cfg, "end_reward", True):                  # ego_forecast.py:126-127 makes the bonus optional
            self.env.end_reward = log.avg_c_reward * cfg.gamma / (1 - cfg.gamma)
        t0 = time.time()
        self.agent.update_params(batch)
        return log, log.sample_time, time.time() - t0, len(batch)

    def save(self, path):
        with to_cpu(*self.nets.values()):
            cp = {k: net.state_dict() for k, net in self.nets.items()}
            cp["running_state"] = self.running_state
            with open(path, "wb") as f:                   # running_state pickles as utils.zfilter.ZFilter
                dump_reference_pickle(cp, f)

    def load(self, path):
        with open(path, "rb") as f:
            cp = load_reference_pickle(f)
        for k, net in self.nets.items():
            net.load_state_dict(cp[k])
        self.running_state = cp["running_state"]
        self.agent.running_state = self.running_state

    def warm_start(self, path, em_cfg=None):
        """ego_forecast.py:60-68: start the policy / value MLPs from an ego_mimic checkpoint; the first affine layer is
        dropped when its input width differs (state LSTM, phase observation or another video width)."""
        from .torch_utils import filter_state_dict
        with open(path, "rb") as f:
            cp = load_reference_pickle(f)
        cfg = self.cfg
        differs = getattr(cfg, "obs_phase", False) or getattr(cfg, "policy_s_net", "id") != "id" or \
            (em_cfg is not None and cfg.policy_v_hdim != em_cfg.policy_v_hdim)
        if differs:
            filter_state_dict(cp["policy_dict"], {"net.affine_layers.0"})
            filter_state_dict(cp["value_dict"], {"net.affine_layers.0"})
        self.policy_net.load_state_dict(cp["policy_dict"], strict=False)
        self.value_net.load_state_dict(cp["value_dict"], strict=False)

    def close(self):
        ro = getattr(self.agent, "_rollout", None)
        if ro is not None:
            ro.drop_prepared()            # a set-up parked by update_params holds the record buffers and the engine
        self.env.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="subject_03")
    ap.add_argument("--data", default=".", help="directory holding config/ and datasets/ (cwd of the reference)")
    ap.add_argument("--synthetic", action="store_true", help="generate a synthetic dataset into --data first")
    ap.add_argument("--num-threads", type=int, default=0)
    ap.add_argument("--num-envs", type=int, default=1024)
    ap.add_argument("--gpu-index", type=int, default=None)
    ap.add_argument("--iter", type=int, default=0)
    ap.add_argument("--max-iter", type=int, default=None)
    ap.add_argument("--dtype", choices=["float32", "float64"], default="float32")
    ap.add_argument("--task", choices=["egomimic", "egoforecast"], default="egomimic")
    args = ap.parse_args()
    rank, world, local = D.init_from_env(args.gpu_index)
    gpu = local if args.gpu_index is None else args.gpu_index
    torch.cuda.set_device(gpu)
    os.makedirs(args.data, exist_ok=True)
    os.chdir(args.data)
    if args.synthetic and rank == 0:
        from .bench_support import write_synthetic_dataset
        write_synthetic_dataset(".", args.cfg, device_index=gpu)
    if world > 1:
        torch.distributed.barrier()
    cfg = (ForecastConfig if args.task == "egoforecast" else Config)(args.cfg, create_dirs=(rank == 0 and args.iter == 0))
    tr = Trainer(cfg, torch.device("cuda", gpu), getattr(torch, args.dtype), num_envs=args.num_envs,
                 num_threads=args.num_threads or None, seed_offset=rank)
    logger = create_logger(os.path.join(cfg.log_dir, "log.txt"), file_handle=rank == 0)
    tb = Logger(cfg.tb_dir) if rank == 0 else None
    if args.iter > 0:
        tr.load("%s/iter_%04d.p" % (cfg.model_dir, args.iter))
    elif args.task == "egoforecast" and cfg.ego_mimic_cfg is not None:
        em_cfg = Config(cfg.ego_mimic_cfg, create_dirs=False)
        cp_path = "%s/iter_%04d.p" % (em_cfg.model_dir, cfg.ego_mimic_iter)
        if os.path.exists(cp_path):
            logger.info("loading model from ego mimic checkpoint: %s" % cp_path)
            tr.warm_start(cp_path, em_cfg)
        else:
            logger.info("no ego mimic checkpoint at %s: training the forecast nets from scratch" % cp_path)
    for i_iter in range(args.iter, args.max_iter or cfg.max_iter_num):
        log, t_s, t_u, _ = tr.iteration(i_iter, cfg.min_batch_size)
        if rank == 0:
            c = log.avg_c_info
            logger.info("%d\tT_sample %.2f\tT_update %.2f\tR_avg %.4f %s\tR_range (%.4f, %.4f)\teps_len_avg %.2f" % (
                i_iter, t_s, t_u, log.avg_c_reward, np.array2string(c, formatter={"all": lambda x: "%.4f" % x}, separator=","),
                log.min_c_reward, log.max_c_reward, log.avg_episode_reward))
            tb.scalar_summary("total_reward", log.avg_c_reward, i_iter)
            tb.scalar_summary("episode_len", log.avg_episode_reward, i_iter)
            for k in range(c.shape[0]):
                tb.scalar_summary("reward_%d" % k, c[k], i_iter)
            if cfg.save_model_interval > 0 and (i_iter + 1) % cfg.save_model_interval == 0:
                tr.save("%s/iter_%04d.p" % (cfg.model_dir, i_iter + 1))
    if rank == 0:
        logger.info("training done!")
    tr.close()


if __name__ == "__main__":
    main()
