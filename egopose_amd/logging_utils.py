"""Console/file logger and scalar-summary writer the training driver expects from ``utils``
(reference: utils/logger.py:5-26, utils/tb_logger.py:24-42, utils/tools.py:55-68).

The reference's ``Logger`` wraps a TensorFlow-1 ``FileWriter``; TensorFlow is not part of this stack, so
scalars are appended to ``<log_dir>/scalars.jsonl`` (one JSON object per call) behind the same method names.
"""
from __future__ import annotations

import json
import logging
import os


def create_logger(filename, file_handle=True):
    logger = logging.getLogger(filename)
    logger.propagate = False
    logger.setLevel(logging.DEBUG)
    if not logger.handlers:
        console = logging.StreamHandler()
        console.setLevel(logging.INFO)
        console.setFormatter(logging.Formatter("%(message)s"))
        logger.addHandler(console)
        if file_handle:
            os.makedirs(os.path.dirname(filename), exist_ok=True)
            fh = logging.FileHandler(filename, mode="a")
            fh.setLevel(logging.DEBUG)
            fh.setFormatter(logging.Formatter("[%(asctime)s] %(message)s"))
            logger.addHandler(fh)
    return logger


class Logger:
    def __init__(self, log_dir, name=None):
        self.name = name
        self.dir = os.path.join(log_dir, name) if name is not None else log_dir
        os.makedirs(self.dir, exist_ok=True)
        self.path = os.path.join(self.dir, "scalars.jsonl")

    def scalar_summary(self, tag, value, step):
        with open(self.path, "a") as f:
            f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")

    def image_summary(self, tag, images, step):
        raise NotImplementedError("image summaries are outside the PPO hot path")

    def histo_summary(self, tag, values, step, bins=1000):
        raise NotImplementedError("histogram summaries are outside the PPO hot path")


def get_body_qposaddr(model):
    """body name -> (first qpos index, one-past-last) from a MuJoCo-like model or a Skeleton."""
    if hasattr(model, "body_qposaddr"):
        return model.body_qposaddr()
    out = {}
    for i, name in enumerate(model.body_names):
        j0 = model.body_jntadr[i]
        if j0 < 0:
            continue
        j1 = j0 + model.body_jntnum[i]
        out[name] = (model.jnt_qposadr[j0], model.jnt_qposadr[j1] if j1 < len(model.jnt_qposadr) else model.nq)
    return out
