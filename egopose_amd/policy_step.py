"""Rollout-time policy step through `egp_policy_gaussian_f32` (csrc/egp_policy.hip): the VideoStateNet concat, the MLP
and the Gaussian head of the reference's `policy_net.select_action` (core/agent.py:38-44, models/policy_gaussian.py:
19-27, models/mlp.py:5-25) for a whole env group in one launch. The module keeps transposed float32 copies of the
weights, packed for the kernel's matrix-core tiles, in persistent buffers (`refresh()` re-reads the live parameters,
addresses stay fixed for hipGraphs)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

_ACT_CODE = {torch.tanh: 0, torch.relu: 1, torch.sigmoid: 2}


def supported(policy_net) -> bool:
    """True when `policy_net` is a PolicyGaussian over a plain MLP with an activation the kernel implements."""
    net = getattr(policy_net, "net", None)
    return (getattr(policy_net, "type", None) == "gaussian" and hasattr(net, "affine_layers")
            and getattr(net, "activation", None) in _ACT_CODE and hasattr(policy_net, "action_mean")
            and len(net.affine_layers) + 1 <= 8 and policy_net.action_mean.weight.dtype == torch.float32
            and max(l.out_features for l in net.affine_layers) <= 2048)


class FusedGaussianPolicy:

    def __init__(self, policy_net, device):
        if not supported(policy_net):
            raise ValueError("policy net is not a float32 PolicyGaussian over an MLP")
        self.lib = L.load()
        self.net = policy_net
        self.layers = list(policy_net.net.affine_layers) + [policy_net.action_mean]
        self.act = _ACT_CODE[policy_net.net.activation]
        # the kernel's packed weight form (include/egopose_hip.h: egp_mlp_layer): one 64-column x 4-feature block per wave load
        # (ONE buffer, layer after layer: the kernel's L2 warm-up walks it as a single range)
        sizes = [int(self.lib.egp_mlp_pack_floats(l.in_features, l.out_features)) for l in self.layers]
        self.wt_all = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
        self.wt = list(torch.split(self.wt_all, sizes))
        self.bias = [torch.empty(l.out_features, dtype=torch.float32, device=device) for l in self.layers]
        self.log_std = torch.empty(self.layers[-1].out_features, dtype=torch.float32, device=device)
        self.desc = (L.MlpLayer * len(self.layers))()
        for i, l in enumerate(self.layers):
            self.desc[i].wt = self.wt[i].data_ptr()
            self.desc[i].bias = self.bias[i].data_ptr()
            self.desc[i].in_dim, self.desc[i].out_dim = l.in_features, l.out_features
        self.in_dim = self.layers[0].in_features
        self.nu = self.layers[-1].out_features
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """Pack the live parameters into the kernel's buffers (call once per rollout, after the optimiser step)."""
        stream = L.current_stream()
        for l, wt, b in zip(self.layers, self.wt, self.bias):
            w = l.weight if l.weight.stride(1) == 1 else l.weight.contiguous()
            L.check(self.lib.egp_mlp_pack_f32(C.c_void_p(w.data_ptr()), int(w.stride(0)), l.in_features, l.out_features,
                                              C.c_void_p(wt.data_ptr()), stream), "egp_mlp_pack_f32")
            b.copy_(l.bias)
        self.log_std.copy_(self.net.action_log_std.reshape(-1))

    def __call__(self, ctx_rows, t_idx, state, action_out, noise=None, mean_out=None):
        """ctx_rows: float32 [n][T][H] (contiguous slab of per-slot context tables), t_idx: int64 [n],
        state: float64 [n][S], noise: float32 [n][nu] or None (mean action), action_out: float64 [n][nu]."""
        n, T, H = ctx_rows.shape
        S = state.shape[1]
        if H + S != self.in_dim:
            raise ValueError("context dim %d + state dim %d != policy input %d" % (H, S, self.in_dim))
        assert ctx_rows.dtype == torch.float32 and ctx_rows.stride(2) == 1 and ctx_rows.stride(1) == H
        assert t_idx.dtype == torch.int64 and t_idx.is_contiguous() and state.dtype == torch.float64 and state.is_contiguous()
        assert action_out.dtype == torch.float64 and action_out.is_contiguous() and action_out.shape == (n, self.nu)
        if noise is not None:
            assert noise.dtype == torch.float32 and noise.is_contiguous() and noise.shape == (n, self.nu)
        if mean_out is not None:
            assert mean_out.dtype == torch.float32 and mean_out.is_contiguous() and mean_out.shape == (n, self.nu)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        L.check(self.lib.egp_policy_gaussian_f32(p(ctx_rows), int(ctx_rows.stride(0)), H, p(t_idx), p(state), S, n, self.desc,
                                                 len(self.layers), self.act, p(self.log_std), p(noise), p(action_out), p(mean_out),
                                                 L.current_stream()), "egp_policy_gaussian_f32")
        return action_out

    def with_filter(self, ctx, ctx_rows, t_idx, qpos, qvel, zf_in, zf_out, clip, y, y2, workspace, action_out, noise=None, mean_out=None,
                    phase_t=None):
        """The filter's apply pass + the policy step in one launch (`egp_policy_gaussian_filter_f32`): the state columns are the
        observations of (qpos, qvel) normalised with `zf_in` merged with the tile statistics `ctx.obs_zfilter_stats` left in
        `workspace`; y / y2 receive them, `zf_out` the merged statistics. `ctx`: the EgpContext of the model."""
        n, T, H = ctx_rows.shape
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        L.check(self.lib.egp_policy_gaussian_filter_f32(ctx.handle, p(ctx_rows), int(ctx_rows.stride(0)), H, p(t_idx), p(qpos), p(qvel),
                                                        p(ctx._phase_t(phase_t, n)), n,
                                                        p(zf_in), p(zf_out), float(clip or 0.0), p(y), p(y2), p(workspace),
                                                        C.cast(self.desc, C.c_void_p), len(self.layers), self.act, p(self.log_std), p(noise),
                                                        p(action_out), p(mean_out), None, None, 0, L.current_stream()),
                "egp_policy_gaussian_filter_f32")
        return action_out
