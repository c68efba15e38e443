"""YAML -> attribute bag for the ego_mimic task (drop-in for ``ego_pose.utils.egomimic_config.Config``).

Behaviour follows /root/reference/ego_pose/utils/egomimic_config.py:9-131: the same attribute names,
defaults, directory layout (``results/egomimic/<id>/{models,results,log,tb}``), data paths
(``datasets/meta/<meta_id>.yml``, ``datasets/features/{expert,cnn_feat}_<id>.p``), PD-gain handling
(``a_ref`` degrees -> radians, gain multipliers) and piecewise-linear adaptive schedules.
The implementation is table driven; YAML files are consumed unmodified. One extension: when
``config/egomimic/<id>.yml`` is not under the cwd, the copy packaged in ``egopose_amd/assets`` is used.
"""
from __future__ import annotations

import os
import shutil

import numpy as np
import yaml

_ASSET_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "config")
_ASSET_CFG = os.path.join(_ASSET_ROOT, "egomimic")

# attribute -> (yaml key, default); attributes whose default depends on another value are handled below
_SCALARS = [
    ("fr_margin", 10), ("state_net_cfg", None), ("state_net_iter", None),
    ("gamma", 0.95), ("tau", 0.95), ("causal", False),
    ("policy_htype", "relu"), ("policy_hsize", [300, 200]), ("policy_v_hdim", 128), ("policy_v_net", "lstm"),
    ("policy_v_net_param", None), ("policy_optimizer", "Adam"), ("policy_lr", 5e-5), ("policy_momentum", 0.0),
    ("policy_weightdecay", 0.0),
    ("value_htype", "relu"), ("value_hsize", [300, 200]), ("value_v_hdim", 128), ("value_v_net", "lstm"),
    ("value_v_net_param", None), ("value_optimizer", "Adam"), ("value_lr", 3e-4), ("value_momentum", 0.0),
    ("value_weightdecay", 0.0),
    ("adv_clip", np.inf), ("clip_epsilon", 0.2), ("log_std", -2.3), ("fix_std", False), ("num_optim_epoch", 10),
    ("min_batch_size", 50000), ("max_iter_num", 1000), ("seed", 1), ("save_model_interval", 100),
    ("reward_id", "quat"), ("reward_weights", None),
    ("env_start_first", False), ("env_init_noise", 0.0), ("env_episode_len", 200), ("obs_type", "full"),
    ("obs_coord", "heading"), ("obs_heading", False), ("obs_vel", "full"), ("root_deheading", True),
    ("sync_exp_interval", 100), ("action_type", "position"),
]


# ego_forecast (ego_pose/utils/egoforecast_config.py:9-131): same table minus the state-net / causal entries, plus the
# ego_mimic warm start, the per-step state nets of VideoForecastNet and a few env switches; other defaults differ
_FORECAST_SCALARS = [(a, d) for a, d in _SCALARS if a not in ("state_net_cfg", "state_net_iter", "causal", "fr_margin")] + [
    ("fr_margin", 10), ("ego_mimic_cfg", None), ("ego_mimic_iter", None),
    ("policy_s_net", "id"), ("policy_s_hdim", None), ("policy_dyn_v", False),
    ("value_s_net", "id"), ("value_s_hdim", None), ("value_dyn_v", False),
    ("end_reward", True), ("obs_phase", False), ("random_cur_t", False),
]


def recreate_dirs(*dirs):
    for d in dirs:
        if os.path.exists(d):
            shutil.rmtree(d)
        os.makedirs(d)


def _schedule(raw, n, fallback):
    arr = np.array(raw if raw is not None else [fallback])
    return np.pad(arr, (0, n - arr.size), "edge")


def joint_and_body_params(cfg):
    """PD tables from the YAML rows [name, k_p, k_d, a_ref(deg), a_scale, torque_limit] and the
    per-body pose weights (egomimic_config.py:108-122)."""
    out = {}
    if "joint_params" in cfg:
        table = np.array([row[1:6] for row in cfg["joint_params"]], dtype=float)
        kp_mul = cfg.get("jkp_multiplier", 1.0)
        out["jkp"] = table[:, 0] * kp_mul
        out["jkd"] = table[:, 1] * cfg.get("jkd_multiplier", kp_mul)
        out["a_ref"] = np.deg2rad(table[:, 2])
        out["a_scale"] = table[:, 3].copy()
        out["torque_lim"] = table[:, 4].copy()
    if "body_params" in cfg:
        out["b_diffw"] = np.array([row[1] for row in cfg["body_params"]])
    return out


def packaged_config(cfg_id):
    with open(os.path.join(_ASSET_CFG, "%s.yml" % cfg_id), "r") as f:
        return yaml.safe_load(f)


class Config:
    task = "egomimic"
    scalars = _SCALARS
    schedules = ("noise_rate", "log_std", "policy_lr")

    def __init__(self, cfg_id=None, create_dirs=False, cfg_dict=None):
        self.id = cfg_id
        if cfg_dict is None:
            path = "config/%s/%s.yml" % (self.task, cfg_id)
            if not os.path.exists(path):
                packaged = os.path.join(_ASSET_ROOT, self.task, "%s.yml" % cfg_id)
                if not os.path.exists(packaged):
                    print("Config file doesn't exist: %s" % path)
                    raise SystemExit(0)
                path = packaged
            with open(path, "r") as f:
                cfg_dict = yaml.safe_load(f)
        cfg = cfg_dict

        # results layout
        self.base_dir = "results"
        self.cfg_dir = "%s/%s/%s" % (self.base_dir, self.task, cfg_id)
        for name in ("model", "result", "log", "tb"):
            setattr(self, name + "_dir", "%s/%s" % (self.cfg_dir, {"model": "models", "result": "results"}.get(name, name)))
        os.makedirs(self.model_dir, exist_ok=True)
        os.makedirs(self.result_dir, exist_ok=True)
        if create_dirs:
            recreate_dirs(self.log_dir, self.tb_dir)

        # data
        self.meta_id = cfg["meta_id"]
        self.data_dir = "datasets"
        with open("%s/meta/%s.yml" % (self.data_dir, self.meta_id), "r") as f:
            self.meta = yaml.safe_load(f)
        self.takes = {split: self.meta[split] for split in ("train", "test")}
        feat = "%s/features/%%s_%%s.p" % self.data_dir
        self.expert_feat_file = feat % ("expert", cfg["expert_feat"]) if "expert_feat" in cfg else None
        self.cnn_feat_file = feat % ("cnn_feat", cfg["cnn_feat"]) if "cnn_feat" in cfg else None

        for attr, default in self.scalars:
            setattr(self, attr, cfg.get(attr, default))
        if getattr(self, "state_net_cfg", None) is not None:
            self.state_net_model = "%s/statereg/%s/models/iter_%04d_inf.p" % (self.base_dir, self.state_net_cfg, self.state_net_iter)

        # adaptive schedules: control points at iterations adp_iter_cp, linear in between
        self.adp_iter_cp = np.array(cfg.get("adp_iter_cp", [0]))
        n_cp = self.adp_iter_cp.size
        self.adp_noise_rate_cp = _schedule(cfg.get("adp_noise_rate_cp"), n_cp, 1.0)
        self.adp_log_std_cp = _schedule(cfg.get("adp_log_std_cp"), n_cp, self.log_std)
        self.adp_policy_lr_cp = _schedule(cfg.get("adp_policy_lr_cp"), n_cp, self.policy_lr)
        self.adp_noise_rate = self.adp_log_std = self.adp_policy_lr = None
        if "init_noise" in self.schedules:
            self.adp_init_noise_cp = _schedule(cfg.get("adp_init_noise_cp"), n_cp, 0.0)
            self.adp_init_noise = None

        # environment / model files
        cwd = os.getcwd()
        self.mujoco_model_file = "%s/assets/mujoco_models/%s.xml" % (cwd, cfg["mujoco_model"])
        self.vis_model_file = "%s/assets/mujoco_models/%s.xml" % (cwd, cfg["vis_model"])

        for name, value in joint_and_body_params(cfg).items():
            setattr(self, name, value)

    def update_adaptive_params(self, i_iter):
        cp = self.adp_iter_cp
        lo = int(np.where(i_iter >= cp)[0][-1])
        hi = lo + int(lo < len(cp) - 1)
        w = (i_iter - cp[lo]) / (cp[hi] - cp[lo]) if hi > lo else 0.0
        for name in self.schedules:
            pts = getattr(self, "adp_%s_cp" % name)
            setattr(self, "adp_" + name, pts[lo] * (1 - w) + pts[hi] * w)


class ForecastConfig(Config):
    """Drop-in for ``ego_pose.utils.egoforecast_config.Config`` (egoforecast_config.py:9-140): results under
    ``results/egoforecast/<id>``, the ego_mimic warm-start ids, the state-net switches of VideoForecastNet, ``end_reward``
    and the extra adaptive schedule ``adp_init_noise`` (:88-95, :131-140)."""
    task = "egoforecast"
    scalars = _FORECAST_SCALARS
    schedules = ("noise_rate", "log_std", "policy_lr", "init_noise")
