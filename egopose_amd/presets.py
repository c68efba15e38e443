"""Model/reward constants of a packaged egomimic config without needing a dataset directory."""
from .config import joint_and_body_params, packaged_config


def config_params(cfg_id="subject_03"):
    cfg = packaged_config(cfg_id)
    out = joint_and_body_params(cfg)
    out["reward_weights"] = dict(cfg.get("reward_weights") or {})
    out["episode_len"] = cfg.get("env_episode_len", 200)
    out["fr_margin"] = cfg.get("fr_margin", 10)
    out["cfg"] = cfg
    return out


def subject_03_params():
    return config_params("subject_03")
