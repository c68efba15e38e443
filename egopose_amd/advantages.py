"""``estimate_advantages`` with the reference's signature (core/common.py:5-25), evaluated by K5 on the GPU."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def estimate_advantages(rewards, masks, values, gamma, tau):
    """rewards (N,), masks (N,), values (N,1) device tensors -> (advantages (N,1), returns (N,1))."""
    if not rewards.is_cuda:
        raise RuntimeError("estimate_advantages runs on the MI355X (HIP kernel K5); got a %s tensor" % rewards.device)
    lib = L.load()
    n = rewards.shape[0]
    sfx = {torch.float64: "f64", torch.float32: "f32"}[rewards.dtype]
    r = rewards.contiguous()
    m = masks.to(r.dtype).contiguous()
    v = values.reshape(-1).to(r.dtype).contiguous()
    adv, ret = torch.empty_like(r), torch.empty_like(r)
    stats = torch.empty(3, dtype=torch.float64, device=r.device)
    ws = torch.empty(int(lib.egp_gae_workspace_bytes(n)), dtype=torch.uint8, device=r.device)
    s = L.current_stream()
    p = lambda t: C.c_void_p(t.data_ptr())
    L.check(getattr(lib, "egp_gae_" + sfx)(p(r), p(m), p(v), n, float(gamma), float(tau), p(adv), p(ret), p(stats), p(ws), s), "egp_gae")
    L.check(getattr(lib, "egp_gae_standardize_" + sfx)(p(adv), n, p(stats), s), "egp_gae_standardize")
    return adv.unsqueeze(1), ret.unsqueeze(1)
