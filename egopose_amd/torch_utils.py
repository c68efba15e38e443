"""Module/device helpers the driver star-imports from ``utils`` (reference: utils/torch.py:1-158)."""
from __future__ import annotations

import contextlib

import numpy as np
import torch

tensor = torch.tensor
DoubleTensor = torch.DoubleTensor
FloatTensor = torch.FloatTensor
LongTensor = torch.LongTensor
ByteTensor = torch.ByteTensor
ones = torch.ones
zeros = torch.zeros


def _device_of(m):
    return m.device if hasattr(m, "device") else next(m.parameters()).device


class _Scoped:
    """Apply a change to some modules now; undo it when used as a context manager and left.

    The reference's helpers take effect at construction (``to_device(device, net)`` is used as a plain
    call at ego_mimic.py:66) and restore on ``__exit__`` -- same here.
    """

    def __init__(self, models, apply, snapshot, restore):
        self.models = [m for m in models if m is not None]
        self._saved = [snapshot(m) for m in self.models]
        self._restore = restore
        for m in self.models:
            apply(m)

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        for m, s in zip(self.models, self._saved):
            self._restore(m, s)
        return False


class to_cpu(_Scoped):
    def __init__(self, *models):
        super().__init__(models, lambda m: m.to(torch.device("cpu")), _device_of, lambda m, d: m.to(d))


class to_device(_Scoped):
    def __init__(self, device, *models):
        super().__init__(models, lambda m: m.to(device), _device_of, lambda m, d: m.to(d))


class to_test(_Scoped):
    def __init__(self, *models):
        super().__init__(models, lambda m: m.train(False), lambda m: m.training, lambda m, t: m.train(t))


class to_train(_Scoped):
    def __init__(self, *models):
        super().__init__(models, lambda m: m.train(True), lambda m: m.training, lambda m, t: m.train(t))


def batch_to(dst, *args):
    return [x.to(dst) for x in args if x is not None]


def get_flat_params_from(models):
    if not hasattr(models, "__iter__"):
        models = (models,)
    return torch.cat([p.data.view(-1) for m in models for p in m.parameters()])


def set_flat_params_to(model, flat_params):
    pos = 0
    for p in model.parameters():
        n = p.numel()
        p.data.copy_(flat_params[pos:pos + n].view_as(p))
        pos += n


def get_flat_grad_from(inputs, grad_grad=False):
    parts = []
    for p in inputs:
        if grad_grad:
            parts.append(p.grad.grad.view(-1))
        elif p.grad is None:
            parts.append(zeros(p.numel(), dtype=p.dtype, device=p.device))
        else:
            parts.append(p.grad.view(-1))
    return torch.cat(parts)


def compute_flat_grad(output, inputs, filter_input_ids=set(), retain_graph=False, create_graph=False):
    inputs = list(inputs)
    wanted = [p for i, p in enumerate(inputs) if i not in filter_input_ids]
    grads = iter(torch.autograd.grad(output, wanted, retain_graph=retain_graph or create_graph, create_graph=create_graph))
    flat = [zeros(p.numel(), dtype=p.dtype, device=p.device) if i in filter_input_ids else next(grads).reshape(-1)
            for i, p in enumerate(inputs)]
    for p in wanted:
        p.grad = None
    return torch.cat(flat)


def set_optimizer_lr(optimizer, lr):
    for group in optimizer.param_groups:
        group["lr"] = lr


def filter_state_dict(state_dict, filter_keys):
    for key in [k for k in state_dict if any(f in k for f in filter_keys)]:
        del state_dict[key]
