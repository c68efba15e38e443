"""Policy / value / video-context networks on PyTorch-ROCm (GEMMs on MFMA via rocBLAS/hipBLASLt/MIOpen).

Parameter names and shapes are those of the reference so checkpoints stay drop-in
(``net.affine_layers.i``, ``action_mean``, ``action_log_std``, ``value_head``, ``v_net.rnn_f/rnn_b``
holding ``nn.LSTMCell`` parameters): models/mlp.py:5-25, models/rnn.py:5-61,
models/video_state_net.py:7-70, core/policy.py:4-23, core/policy_gaussian.py:7-38, core/critic.py:5-18,
core/distributions.py:6-25 of /root/reference.

What differs is how they run: the temporal net consumes whole batches of windows -- (T, B, D) with B =
all envs that reset this tick / all episodes of the PPO batch -- and the bi-LSTM is evaluated with one
input-projection GEMM per direction plus a fused recurrent sweep (MIOpen ``lstm`` when the dtype allows,
otherwise a cell loop over pre-projected gates) instead of 2*T LSTMCell calls at batch 1.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

import os

from . import dist as _dist
from . import gemm as _gemm
from . import gemm_tuning as _tuning
from . import lstm as _hip_lstm

# Length buckets of the train-mode forward LSTM (1 = off: measured SLOWER on the MI355X -- the persistent LSTM kernel is
# bound by its per-timestep latency, not by the batch, so four short launches in a row cost more than one long one; the
# code path stays for throughput-bound shapes, a module attribute, no switch). Running the two directions on two HIP streams
# was also tried: no gain in the update, and an intermittent stall next to the resident K1 streams -- removed.
_FWD_BUCKETS = 1
_LSTM_IMPL = os.environ.get("EGP_LSTM", "hip")      # "torch" forces the MIOpen / cell-loop paths (A/B runs)

_ACT = {"tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid}


def bucket_rows(x):
    """Zero-pad a big (N, D) batch to the next row bucket so its GEMM shapes come from a small tuned set
    (gemm_tuning.py). Returns (x_padded, N) or (x, None)."""
    if _tuning.enabled() and x.is_cuda and x.dim() == 2 and x.shape[0] >= 4 * _tuning.ROW_BUCKET:
        pad = _tuning.pad_to(x.shape[0], _tuning.ROW_BUCKET)
        if pad:
            return nn.functional.pad(x, (0, 0, 0, pad)), x.shape[0]
    return x, None


def _take_lengths(take_offset, n_rows):
    off = np.asarray(take_offset, dtype=np.int64).ravel()
    return np.diff(np.concatenate((off, [int(n_rows)])))


def _check_windows(take_len, expert_ind, start_ind, before, after):
    """Every window [start - before, start + after) inside its take, else ValueError (host-side, numpy indices)."""
    e = np.asarray(expert_ind, dtype=np.int64).ravel()
    s = np.asarray(start_ind, dtype=np.int64).ravel()
    if e.size == 0:
        return
    if e.min() < 0 or e.max() >= len(take_len):
        raise ValueError("take index out of range: [%d, %d] with %d takes" % (e.min(), e.max(), len(take_len)))
    bad = (s - before < 0) | (s + after > take_len[e])
    if bad.any():
        k = int(np.nonzero(bad)[0][0])
        raise ValueError("CNN-feature window [%d, %d) leaves take %d (%d frames): %d of %d windows do"
                         % (s[k] - before, s[k] + after, e[k], take_len[e[k]], int(bad.sum()), e.size))


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dims=(128, 128), activation="tanh"):
        super().__init__()
        self.activation = _ACT[activation]
        dims = [input_dim] + list(hidden_dims)
        self.out_dim = dims[-1]
        self.affine_layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for layer in self.affine_layers:
            x = self.activation(layer(x))
        return x


class _DiagLogProb(torch.autograd.Function):
    """sum_j log N(value_j; loc_j, exp(log_std_j)) per row (core/distributions.py: normal_log_density + sum(1, keepdim)) in four
    full-size element-wise kernels forward and two backward (torch.distributions.Normal.log_prob takes about eight each way,
    plus a validity check of the sample that synchronises with the host). log_std: one row (1, d), broadcast over the batch."""

    @staticmethod
    def forward(ctx, value, loc, log_std):
        inv_std = torch.exp(-log_std)
        z = (value - loc) * inv_std
        ctx.save_for_backward(z, inv_std)
        const = log_std.sum() + 0.5 * log_std.shape[-1] * math.log(2.0 * math.pi)
        return (z * z).sum(1, keepdim=True).mul_(-0.5).sub_(const)

    @staticmethod
    def backward(ctx, g):
        z, inv_std = ctx.saved_tensors
        d_loc = d_ls = None
        if ctx.needs_input_grad[1]:
            d_loc = (z * inv_std).mul_(g)
        if ctx.needs_input_grad[2]:
            d_ls = ((z * z - 1.0) * g).sum(0, keepdim=True)
        d_val = None
        if ctx.needs_input_grad[0]:
            d_val = -((z * inv_std) * g)
        return d_val, d_loc, d_ls


class DiagGaussian(torch.distributions.Normal):
    """Normal with summed log-prob and the reference's 'KL to a detached copy of itself'. `log_std_row`: the (1, d) log
    standard deviation the scale was expanded from, when the caller has it (PolicyGaussian): log_prob then takes the short
    route (_DiagLogProb)."""

    def __init__(self, loc, scale, log_std_row=None):
        super().__init__(loc, scale, validate_args=False)       # (the argument checks cost a device reduction + host sync per call)
        self._log_std_row = log_std_row

    def kl(self):
        mu0, s0 = self.loc.detach(), self.scale.detach()
        ls1 = self.scale.log()
        out = ls1 - ls1.detach() + (s0.pow(2) + (mu0 - self.loc).pow(2)) / (2.0 * self.scale.pow(2)) - 0.5
        return out.sum(1, keepdim=True)

    def log_prob(self, value):
        if self._log_std_row is not None and value.dim() == 2:
            return _DiagLogProb.apply(value, self.loc, self._log_std_row)
        return super().log_prob(value).sum(1, keepdim=True)

    def mean_sample(self):
        return self.loc


class Policy(nn.Module):
    def select_action(self, x, mean_action=False):
        dist = self.forward(x)
        return dist.mean_sample() if mean_action else dist.sample()

    def get_kl(self, x):
        return self.forward(x).kl()

    def get_log_prob(self, x, action):
        return self.forward(x).log_prob(action)


class PolicyGaussian(Policy):
    def __init__(self, net, action_dim, net_out_dim=None, log_std=0, fix_std=False):
        super().__init__()
        self.type = "gaussian"
        self.net = net
        self.action_mean = nn.Linear(net.out_dim if net_out_dim is None else net_out_dim, action_dim)
        with torch.no_grad():
            self.action_mean.weight.mul_(0.1)
            self.action_mean.bias.zero_()
        self.action_log_std = nn.Parameter(torch.full((1, action_dim), float(log_std)), requires_grad=not fix_std)

    def mean_std(self, x):
        if _gemm.mlp_head_available(x, getattr(self.net, "affine_layers", ()), self.action_mean, getattr(self.net, "activation", None)):
            # the whole head(MLP(x)) as one autograd node on the split-operand bf16 GEMM kernel (gemm.py); `_egp_ctx_cols`
            # (a tag VideoStateNet.forward leaves on the tensor it returns, not a module attribute): only the leading
            # video-context columns of that input carry a gradient -- any other input gets the full gradient
            # (set by the agent): only the leading video-context columns of the input carry a gradient
            if isinstance(x, _gemm.GatheredInput):      # [context | state] never formed: the first layer gathers it itself
                mean = _gemm.gather_mlp_head(x, self.net.affine_layers, self.action_mean)
            else:
                mean = _gemm.mlp_head(x, self.net.affine_layers, self.action_mean, getattr(x, "_egp_ctx_cols", None))
        else:
            if isinstance(x, _gemm.GatheredInput):
                x = x.materialize()
            x, n = bucket_rows(x)
            mean = self.action_mean(self.net(x))
            if n is not None:
                mean = mean[:n]
        return mean, torch.exp(self.action_log_std).expand_as(mean)         # (a view: no (n, d) tensor of standard deviations)

    def forward(self, x):
        mean, std = self.mean_std(x)
        return DiagGaussian(mean, std, self.action_log_std if self.action_log_std.dim() == 2 and self.action_log_std.shape[0] == 1 else None)

    def get_fim(self, x):
        mean, _ = self.mean_std(x)
        cov_inv = self.action_log_std.exp().pow(-2).squeeze(0).repeat(x.size(0))
        offset, std_id, std_index = 0, 0, 0
        for i, (name, p) in enumerate(self.named_parameters()):
            if name == "action_log_std":
                std_id, std_index = i, offset
            offset += p.numel()
        return cov_inv.detach(), mean, {"std_id": std_id, "std_index": std_index}


class Value(nn.Module):
    def __init__(self, net, net_out_dim=None):
        super().__init__()
        self.net = net
        self.value_head = nn.Linear(net.out_dim if net_out_dim is None else net_out_dim, 1)
        with torch.no_grad():
            self.value_head.weight.mul_(0.1)
            self.value_head.bias.zero_()

    def forward(self, x):
        if _gemm.mlp_head_available(x, getattr(self.net, "affine_layers", ()), self.value_head, getattr(self.net, "activation", None)):
            if isinstance(x, _gemm.GatheredInput):
                return _gemm.gather_mlp_head(x, self.net.affine_layers, self.value_head)
            return _gemm.mlp_head(x, self.net.affine_layers, self.value_head, getattr(x, "_egp_ctx_cols", None))
        if isinstance(x, _gemm.GatheredInput):
            x = x.materialize()
        x, n = bucket_rows(x)
        out = self.value_head(self.net(x))
        return out if n is None else out[:n]


# ------------------------------------------------------------------------------------------------ temporal nets
def _lstm_sweep(cell: nn.LSTMCell, x, reverse):
    """(T,B,D) -> (T,B,H) for one direction; zero initial state."""
    T, B, _ = x.shape
    H = cell.hidden_size
    if _LSTM_IMPL != "torch" and _hip_lstm.available(x, cell):
        return _hip_lstm.lstm_direction(cell, x.contiguous(), reverse)      # persistent HIP recurrence
    fused_ok = x.is_cuda and x.dtype in (torch.float32, torch.float16, torch.bfloat16)
    if fused_ok:
        h0 = x.new_zeros(1, B, H)
        xs = x.flip(0) if reverse else x
        out, _, _ = torch._VF.lstm(xs, (h0, h0), [cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh],
                                   True, 1, 0.0, cell.training, False, False)
        return out.flip(0) if reverse else out
    # generic path: one GEMM for every timestep's input projection, then the recurrence
    gates_x = torch.addmm(cell.bias_ih + cell.bias_hh, x.reshape(T * B, -1), cell.weight_ih.t()).view(T, B, 4 * H)
    w_hh_t = cell.weight_hh.t()
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        g = torch.addmm(gates_x[t], h, w_hh_t)
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, 0)


class RNN(nn.Module):
    def __init__(self, input_dim, out_dim, cell_type="lstm", bi_dir=False):
        super().__init__()
        self.input_dim, self.out_dim, self.cell_type, self.bi_dir = input_dim, out_dim, cell_type, bi_dir
        self.mode = "batch"
        make = nn.LSTMCell if cell_type == "lstm" else nn.GRUCell
        hidden = out_dim // 2 if bi_dir else out_dim
        self.rnn_f = make(input_dim, hidden)
        if bi_dir:
            self.rnn_b = make(input_dim, hidden)
        self.hx = self.cx = None
        self.ragged = None       # set by the caller around a batch forward: (order, steps) of lstm.ragged_order

    def set_mode(self, mode):
        self.mode = mode

    def initialize(self, batch_size=1):
        if self.mode == "step":
            p = self.rnn_f.weight_hh
            self.hx = p.new_zeros(batch_size, self.rnn_f.hidden_size)
            self.cx = p.new_zeros(batch_size, self.rnn_f.hidden_size) if self.cell_type == "lstm" else None

    def _sweep(self, cell, x, reverse):
        if self.cell_type == "lstm":
            return _lstm_sweep(cell, x, reverse)
        h = x.new_zeros(x.size(1), cell.hidden_size)
        outs = [None] * x.size(0)
        for t in (range(x.size(0) - 1, -1, -1) if reverse else range(x.size(0))):
            h = cell(x[t], h)
            outs[t] = h
        return torch.stack(outs, 0)

    def forward(self, x):
        if self.mode == "step":
            self.hx = self.hx.to(x.device)
            if self.cell_type == "lstm":
                self.hx, self.cx = self.rnn_f(x, (self.hx, self.cx.to(x.device)))
            else:
                self.hx = self.rnn_f(x, self.hx)
            return self.hx
        if (self.bi_dir and self.cell_type == "lstm" and _LSTM_IMPL != "torch"
                and _hip_lstm.group_available(x, [self.rnn_f, self.rnn_b])):
            # both directions in one grouped launch each way, writing the halves of one (T, B, 2H) buffer (lstm.LstmGroup)
            return _hip_lstm.lstm_group(x, [self.rnn_f, self.rnn_b], [False, True], pairs=True, ragged=self.ragged)[0]
        out = self._sweep(self.rnn_f, x, False)
        if self.bi_dir:
            out = torch.cat((out, self._sweep(self.rnn_b, x, True)), 2)
        return out


class VideoStateNet(nn.Module):
    """Temporal net over precomputed CNN features, concatenated in front of the state.

    test mode: ``initialize(window)`` with window (T+2m, D) for ONE episode, or (T+2m, B, D) for a batch of
    episodes (the lockstep rollout), keeps ``v_out`` = net(window)[m:-m]; ``forward(state)`` prepends
    v_out[t] and advances t (single-episode form) -- the batched rollout indexes ``v_out`` itself.
    train mode: ``initialize((masks, cnn_feat, v_metas))`` segments the flat batch into episodes and builds
    the padded context (max_len+2m, n_ep, D) and flat gather indices; ``forward(states)`` runs the net over
    the context and gathers one row per sample.
    """

    def __init__(self, cnn_feat_dim, v_hdim=128, v_margin=10, v_net_type="lstm", v_net_param=None, causal=False):
        super().__init__()
        if v_net_type != "lstm":
            raise NotImplementedError("only the 'lstm' video net is on the hot path (tcn is out of scope)")
        self.mode = "test"
        self.cnn_feat_dim, self.v_hdim, self.v_margin, self.v_net_type = cnn_feat_dim, v_hdim, v_margin, v_net_type
        self.v_net = RNN(cnn_feat_dim, v_hdim, v_net_type, bi_dir=not causal)
        self.v_out = None
        self.t = 0
        self.indices = None
        self.gather_indices = None
        self._gather_tm = None
        self._gather_unique = False
        self.cnn_feat_ctx = None
        self._buckets = None
        self._v_ctx = None
        self._ctx_key = None
        self._cnn_table = None   # optional (device table, take offsets) installed by the env
        self._ragged = None      # (order, steps) of the train-mode windows for the grouped HIP sweeps

    def set_mode(self, mode):
        self.mode = mode

    def forward_v_net(self, x):
        return self.v_net(x)

    def attach_feature_table(self, table, take_offset):
        """Device-resident concatenation of all takes' features (+ row offsets) for gather-built contexts."""
        host = np.asarray(take_offset, dtype=np.int64)
        cur = getattr(self, "_cnn_table", None)
        if cur is not None and cur[0] is table and np.array_equal(getattr(self, "_take_offset_host", None), host):
            return                     # (same table as last time: no upload -- a copy from pageable memory blocks the host until the stream drains)
        self._cnn_table = (table, torch.as_tensor(host, dtype=torch.long, device=table.device))
        self._take_offset_host = host.copy()
        self._take_len = _take_lengths(take_offset, table.shape[0])

    def check_windows(self, expert_ind, start_ind, length):
        """Rows [start - m, start + length + m) must lie inside their own take: the table is the concatenation of all
        takes, so a window that leaves its take would silently read a neighbour's frames (the reference's numpy slice
        comes up short and raises on assignment, models/video_state_net.py:52-55)."""
        _check_windows(self._take_len, expert_ind, start_ind, self.v_margin, length + self.v_margin)

    def window_features(self, expert_ind, start_ind, length):
        """Gather windows [start-m, start+length+m) of the given takes -> (length+2m, B, D) on device."""
        table, off = self._cnn_table
        m = self.v_margin
        base = off[expert_ind.long()] + start_ind.long() - m
        rows = base.unsqueeze(0) + torch.arange(length + 2 * m, device=table.device).unsqueeze(1)
        return table[rows]

    def initialize(self, x):
        m = self.v_margin
        if self.mode == "test":
            p = next(self.parameters())
            x = x.to(device=p.device, dtype=p.dtype)
            single = x.dim() == 2
            out = self.forward_v_net(x.unsqueeze(1) if single else x)[m:-m]
            self.v_out = out.squeeze(1) if single else out
            self.t = 0
            return
        masks, cnn_feat, v_metas = x
        device, dtype = masks.device, masks.dtype
        self._frames = None
        ends = torch.nonzero(masks == 0).flatten().cpu().numpy()
        n = masks.shape[0]
        starts = np.concatenate(([0], ends[:-1] + 1))
        lens = ends - starts + 1
        # the padded window length is a batch-wide quantity in the reference (one process, one batch):
        # keep it global when the batch is sharded over ranks
        max_len = _dist.global_max(int(lens.max()), device)
        idx = np.arange(n)
        ep_of = np.repeat(np.arange(len(ends)), lens)
        covered = int(ends[-1]) + 1
        idx[:covered] = ep_of * max_len + (np.arange(covered) - np.repeat(starts, lens))
        self.indices = idx
        meta = np.asarray(v_metas)[ends]
        if self._cnn_table is not None and self._cnn_table[0].device == device:
            # episode count rounded up to a bucket (gemm_tuning.py): the extra windows repeat episode 0, nothing
            # gathers from them, so their gradient contribution is exactly zero
            pad = _tuning.pad_to(len(ends), _tuning.EPISODE_BUCKET) if _tuning.enabled() and len(ends) >= 4 * _tuning.EPISODE_BUCKET else 0
            if pad:
                meta = np.concatenate((meta, np.repeat(meta[:1], pad, axis=0)), 0)
            self.check_windows(meta[:, 0], meta[:, 1], max_len)
            e_ind = torch.as_tensor(meta[:, 0], device=device)
            s_ind = torch.as_tensor(meta[:, 1], device=device)
            self.cnn_feat_ctx = self.window_features(e_ind, s_ind, max_len).to(dtype)
            # The windows are runs of consecutive rows of the feature table: window b starts at row base[b]. When the table is
            # (much) smaller than the windows laid end to end -- the takes' frames are visited by many episodes -- the LSTMs'
            # input projection is computed per frame of the table instead of per window row (lstm.LstmGroup, `frames`).
            table = self._cnn_table[0]
            n_rows = self.cnn_feat_ctx.shape[0] * self.cnn_feat_ctx.shape[1]
            if table.dtype == dtype and table.is_contiguous() and 2 * table.shape[0] <= n_rows:
                base = self._take_offset_host[meta[:, 0].astype(np.int64)] + meta[:, 1].astype(np.int64) - m
                self._frames = (table, torch.as_tensor(base.astype(np.int32), device=device))
        else:
            ctx = np.zeros((max_len + 2 * m, len(ends), self.cnn_feat_dim))
            for e, (ei, si) in enumerate(meta):
                ctx[:, e, :] = cnn_feat[int(ei)][int(si) - m: int(si) + max_len + m]
            self.cnn_feat_ctx = torch.as_tensor(ctx, dtype=dtype, device=device)
        self.gather_indices = torch.as_tensor(idx, dtype=torch.long, device=device)
        # the same rows addressed in the net's own (time, episode) order: forward() gathers straight from the LSTM output
        # instead of slicing off the margins, transposing and copying it first
        tm = ((idx % max_len) + m) * self.cnn_feat_ctx.shape[1] + idx // max_len
        self._gather_tm = torch.as_tensor(tm, dtype=torch.long, device=device)
        # the backward pass's scatter relies on distinct rows. Samples inside episodes have them by construction (episode e,
        # offset < len_e <= max_len -> a distinct (frame, episode) cell); only a batch with samples after its last episode
        # end (idx = the sample's own number there) needs the look
        self._gather_unique = covered == n or np.unique(tm).size == tm.size
        self._ctx_key = (int(max_len), meta.shape[0], hash(meta.tobytes()))       # which windows cnn_feat_ctx holds
        # Ragged sweeps (lstm.ragged_order): the forward direction's output at frame t depends on frames <= t only and only
        # frames [m, m + len_e) of an episode are ever gathered, so it stops after m + len_e steps (workgroups of sequences
        # sorted by length); the backward direction starts at the end of the padded window and runs it all, as in the
        # reference. Windows added to fill an episode bucket need no step at all.
        steps = np.zeros(self.cnn_feat_ctx.shape[1], np.int64)
        steps[:len(lens)] = lens + m
        self._ragged = _hip_lstm.ragged_order(steps, device, T=self.cnn_feat_ctx.shape[0]) if self.cnn_feat_ctx.is_cuda else None
        # Length buckets for the forward direction: its output at frame t only depends on frames <= t and only frames
        # [m, m + len_e) of an episode are ever gathered, so episodes sorted by length let the forward LSTM stop early
        # (the backward direction starts at the end of the padded window and must run it all, as in the reference).
        self._buckets = None
        n_ep = len(ends)
        if _FWD_BUCKETS > 1 and n_ep >= 4 * _FWD_BUCKETS and self.v_net_type == "lstm":
            order = np.argsort(-lens, kind="stable")
            rank = np.empty(n_ep, np.int64)
            rank[order] = np.arange(n_ep)
            idx_s = np.arange(n)
            idx_s[:covered] = rank[ep_of] * max_len + (np.arange(covered) - np.repeat(starts, lens))
            self._gather_sorted = torch.as_tensor(idx_s, dtype=torch.long, device=device)
            self._ctx_sorted = self.cnn_feat_ctx.index_select(1, torch.as_tensor(order, device=device))
            cuts = [n_ep * k // _FWD_BUCKETS for k in range(_FWD_BUCKETS + 1)]
            lens_sorted = lens[order]
            self._buckets = [(cuts[k], cuts[k + 1], int(m + lens_sorted[cuts[k]])) for k in range(_FWD_BUCKETS)]

    _TRAIN_CONTEXT = ("indices", "cnn_feat_ctx", "gather_indices", "_gather_tm", "_gather_unique", "_ctx_key", "_ragged",
                      "_buckets", "_gather_sorted", "_ctx_sorted", "_frames")

    def adopt_train_context(self, other, x):
        """``initialize(x)`` in train mode when ``other`` has just been initialised with the SAME ``x``: the episode
        segmentation, the gather indices, the gathered feature windows and the ragged order depend on ``x``, the margin and
        the feature table only (not on the net's weights), so the policy's and the value function's front ends
        (ego_pose/core/agent_ego.py:34-41 initialises one after the other) share them -- one device sync and one pass of
        index arithmetic over the batch instead of two. Falls back to ``initialize`` when the two nets are not alike."""
        alike = (isinstance(other, VideoStateNet) and self.mode == "train" and other.mode == "train" and other.cnn_feat_ctx is not None
                 and other._ctx_key is not None and self.v_margin == other.v_margin and self.cnn_feat_dim == other.cnn_feat_dim
                 and self.v_net_type == other.v_net_type and self._cnn_table is not None and other._cnn_table is not None
                 and self._cnn_table[0].data_ptr() == other._cnn_table[0].data_ptr()
                 and np.array_equal(self._take_offset_host, other._take_offset_host) and other.cnn_feat_ctx.dtype == x[0].dtype)
        if not alike:
            return self.initialize(x)
        for k in self._TRAIN_CONTEXT:
            if hasattr(other, k):
                setattr(self, k, getattr(other, k))

    def _bucketed_context(self):
        """(T - 2m, n_ep sorted by length, v_hdim): backward direction over the full window, forward direction per length bucket."""
        ctx, rnn = self._ctx_sorted, self.v_net
        T, B, _ = ctx.shape
        Hd = rnn.rnn_f.hidden_size
        out_f = ctx.new_zeros(T, B, Hd)
        for i0, i1, Tb in self._buckets:
            Tb = min(Tb, T)
            out_f[:Tb, i0:i1] = rnn._sweep(rnn.rnn_f, ctx[:Tb, i0:i1], False)
        if not rnn.bi_dir:
            return out_f
        return torch.cat((out_f, rnn._sweep(rnn.rnn_b, ctx, True)), 2)

    def forward(self, x, lazy_width=0):
        """`lazy_width` > 0 (train mode; the width of the consumer's first layer, passed per call by an agent whose heads can
        read the parts themselves): the result may be a gemm.GatheredInput -- [context rows | x] not yet formed -- instead of
        a tensor. The tensors this returns in train mode carry `_egp_ctx_cols` = v_hdim when x needs no gradient: their
        trailing columns are raw states, so a consumer may skip that part of the input gradient."""
        if self.mode == "test":
            out = torch.cat((self.v_out[[self.t], :], x), dim=1)
            self.t += 1
            return out
        m = self.v_margin
        def tagged(out):
            if not x.requires_grad:
                out._egp_ctx_cols = self.v_hdim
            return out
        if self._buckets is not None:
            ctx = self._bucketed_context()[m:-m].transpose(0, 1).reshape(-1, self.v_hdim)
            return tagged(torch.cat((ctx.index_select(0, self._gather_sorted), x), dim=1))
        ctx = None
        if self._v_ctx is not None:          # computed together with another net's (grouped_video_context)
            (ctx, with_grad), self._v_ctx = self._v_ctx, None
            if with_grad != torch.is_grad_enabled():
                ctx = None                   # left over from a pass in the other autograd mode: never reuse it
        if ctx is None:
            self.v_net.ragged = self._ragged
            try:
                ctx = self.forward_v_net(self.cnn_feat_ctx)
            finally:
                self.v_net.ragged = None
        ctx2d = ctx.reshape(-1, self.v_hdim)
        if self._gather_unique and _gemm.gather_concat_available(ctx2d, self._gather_tm, x):
            if lazy_width and _gemm.fused_gather_available(self.v_hdim, lazy_width, x.shape[1]):
                return _gemm.GatheredInput(ctx2d, self._gather_tm, x)            # the consumer's first layer gathers (gemm.GatherMlpHead)
            return tagged(_gemm.GatherConcat.apply(ctx2d, self._gather_tm, x))  # gather + concatenation in one pass
        return tagged(torch.cat((ctx2d.index_select(0, self._gather_tm), x), dim=1))


class Bf16Shadow:
    """bfloat16 compute copy of a convolutional encoder whose float32 parameters stay the MASTER weights (state_dict,
    optimizer). `torch.autocast` is not the way to bf16 on this stack: it re-casts weights and activations around every
    op (127 ms per 256-frame ResNet-18 step against 37 ms in float32); a module that simply IS bf16 runs MIOpen's
    implicit-GEMM MFMA convolutions end to end (21 ms). Normalisation layers are shared with the master (float32 affine
    parameters and running statistics, mixed-dtype batch norm), convolution / linear weights are bf16 copies:
    `pull()` after an optimizer step or a checkpoint load, `push_grads()` between backward and the optimizer step."""

    def __init__(self, master):
        import copy
        self.master = master
        self.shadow = copy.deepcopy(master)
        for name, m in list(master.named_modules()):
            if isinstance(m, nn.modules.batchnorm._BatchNorm):
                parent = self.shadow
                *path, leaf = name.split(".")
                for part in path:
                    parent = getattr(parent, part)
                setattr(parent, leaf, m)
        mp = dict(master.named_parameters())
        self.pairs = []
        for n, p in self.shadow.named_parameters():
            if p is not mp[n]:
                p.data = p.data.to(torch.bfloat16)
                self.pairs.append((mp[n], p))

    @torch.no_grad()
    def pull(self):
        torch._foreach_copy_([s for _, s in self.pairs], [m for m, _ in self.pairs])

    @torch.no_grad()
    def push_grads(self):
        dst, src = [], []
        for m, s in self.pairs:
            if s.grad is None:
                continue
            if m.grad is None:
                m.grad = torch.empty_like(m)
            dst.append(m.grad)
            src.append(s.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        for _, s in self.pairs:
            s.grad = None

    def __call__(self, x):
        return self.shadow(x.to(torch.bfloat16)).float()


class VideoRegNet(nn.Module):          # (ResNet is defined further down; resolved at construction time)
    """State regressor of models/video_reg_net.py:10-59: [ResNet-18 per frame ->] temporal net -> MLP (relu) -> Linear.
    `no_cnn=True` (what the evaluation drivers load, ego_pose/ego_mimic_eval.py:71-80) takes precomputed features
    x: (T, B, cnn_fdim); with the encoder x: (T, B, 3, H, W). Output (T*B, out_dim)."""

    def __init__(self, out_dim, v_hdim, cnn_fdim, no_cnn=True, frame_shape=(3, 224, 224), mlp_dim=(300, 200),
                 cnn_type="resnet", v_net_type="lstm", v_net_param=None, causal=False):
        super().__init__()
        if not no_cnn and cnn_type != "resnet":
            raise NotImplementedError("only the 'resnet' image encoder is implemented (mobile net is out of scope)")
        if v_net_type != "lstm":
            raise NotImplementedError("only the 'lstm' video net is implemented (tcn is out of scope)")
        self.out_dim, self.cnn_fdim, self.v_hdim, self.no_cnn = out_dim, cnn_fdim, v_hdim, no_cnn
        self.frame_shape = tuple(frame_shape)
        self.cnn = None if no_cnn else ResNet(cnn_fdim)
        self.v_net_type = v_net_type
        self.v_net = RNN(cnn_fdim, v_hdim, v_net_type, bi_dir=not causal)
        self.mlp = MLP(v_hdim, mlp_dim, "relu")
        self.linear = nn.Linear(self.mlp.out_dim, out_dim)

    def forward_v_net(self, x):
        return self.v_net(x)

    def channels_last(self):
        """Keep the encoder's weights and activations NHWC (MIOpen's fp32 kernels run the ResNet-18 step 22 % faster in
        that layout on the MI355X: 46.1 -> 37.7 ms per 256-frame clip; same arithmetic). Returns self."""
        if self.cnn is not None:
            self.cnn.to(memory_format=torch.channels_last)
            self._nhwc = True
        return self

    def bf16_encoder(self, on=True):
        """Run the image encoder in bfloat16 on the matrix cores with its float32 parameters as master weights
        (BASELINE config 4; `Bf16Shadow`). Training loops call `encoder_grads_ready()` after backward and
        `encoder_stepped()` after the optimizer step (or a checkpoint load). Returns self."""
        self.__dict__["_enc16"] = Bf16Shadow(self.cnn) if (on and self.cnn is not None) else None
        return self

    def encoder_grads_ready(self):
        if self.__dict__.get("_enc16") is not None:
            self._enc16.push_grads()

    def encoder_stepped(self):
        if self.__dict__.get("_enc16") is not None:
            self._enc16.pull()

    def _encode(self, frames):
        # The bf16 copy serves OPTIMISATION steps only (train mode with autograd on). Everything whose output is kept --
        # the features gen_cnn_feature stores for the RL nets, test-mode predictions, eval metrics -- goes through the
        # float32 master encoder: ~8 mantissa bits would change stored features with no opt-in.
        enc = self.__dict__.get("_enc16")
        if enc is not None and self.training and torch.is_grad_enabled():
            enc.shadow.train(True)
            return enc(frames)
        return self.cnn(frames)

    def _frames(self, x):
        x = x.reshape((-1,) + self.frame_shape)
        return x.contiguous(memory_format=torch.channels_last) if getattr(self, "_nhwc", False) else x

    def forward(self, x):
        if self.cnn is not None:        # x: (T, B, 3, H, W) optical-flow frames -> per-frame features
            x = self._encode(self._frames(x)).view(-1, x.size(1), self.cnn_fdim)
        x = self.forward_v_net(x).reshape(-1, self.v_hdim)
        return self.linear(self.mlp(x))

    def get_cnn_feature(self, x):
        return self._encode(self._frames(x))


class VideoForecastNet(nn.Module):
    """Policy/value front end of ego_forecast (models/video_forecast_net.py:7-111): the video net sees only the
    `v_margin` frames BEFORE the episode (causal LSTM, last output kept for the whole episode), the state goes through
    its own LSTM (`s_net_type='lstm'`, stepped one env-step at a time while sampling) or through unchanged ('id').

    test mode   ``initialize(window)``: window (>= v_margin, D) of ONE episode, or (v_margin, B, D) for a batch of
                episodes (the lockstep rollout); ``forward(state)`` -> cat(v_out, s_net(state)). The batched rollout
                steps the state LSTM itself through ``s_step``.
    train mode  ``initialize((masks, cnn_feat, v_metas))`` segments the flat batch into episodes;
                ``forward(states)`` scatters the states into a padded (max_len, n_ep, state_dim) block, runs both
                nets and gathers one row per sample.
    """

    def __init__(self, cnn_feat_dim, state_dim, v_hdim=128, v_margin=10, v_net_type="lstm", v_net_param=None,
                 s_hdim=None, s_net_type="id", dynamic_v=False):
        super().__init__()
        if v_net_type != "lstm":
            raise NotImplementedError("only the 'lstm' video net is implemented (tcn is out of scope)")
        if s_net_type not in ("id", "lstm"):
            raise ValueError("s_net_type must be 'id' or 'lstm'")
        s_hdim = state_dim if s_hdim is None else s_hdim
        self.cnn_feat_dim, self.state_dim, self.v_hdim, self.v_margin = cnn_feat_dim, state_dim, v_hdim, v_margin
        self.v_net_type, self.s_net_type, self.s_hdim, self.dynamic_v = v_net_type, s_net_type, s_hdim, dynamic_v
        self.out_dim = v_hdim + s_hdim
        self.v_net = RNN(cnn_feat_dim, v_hdim, v_net_type, bi_dir=False)
        if s_net_type == "lstm":
            self.s_net = RNN(state_dim, s_hdim, s_net_type, bi_dir=False)
        self.v_out = None
        self.t = 0
        self.indices = self.gather_indices = self.cnn_feat_ctx = None
        self.num_episode = self.max_episode_len = None
        self._cnn_table = None
        self._grp = self._ctx_key = None
        self.set_mode("test")

    def set_mode(self, mode):
        self.mode = mode
        if self.s_net_type == "lstm":
            self.s_net.set_mode("batch" if mode == "train" else "step")

    def forward_v_net(self, x):
        return self.v_net(x)

    def attach_feature_table(self, table, take_offset):
        host = np.asarray(take_offset, dtype=np.int64)
        cur = getattr(self, "_cnn_table", None)
        if cur is not None and cur[0] is table and np.array_equal(getattr(self, "_take_offset_host", None), host):
            return                     # (same table as last time: no blocking upload)
        self._cnn_table = (table, torch.as_tensor(host, dtype=torch.long, device=table.device))
        self._take_offset_host = host.copy()
        self._take_len = _take_lengths(take_offset, table.shape[0])

    def check_windows(self, expert_ind, start_ind, length=0):
        """Rows [start - v_margin, start + length) must lie inside their own take (see VideoStateNet.check_windows)."""
        _check_windows(self._take_len, expert_ind, start_ind, self.v_margin, length)

    def window_features(self, expert_ind, start_ind, length=0):
        """Rows [start - v_margin, start + length) of the given takes -> (v_margin + length, B, D) on device."""
        table, off = self._cnn_table
        base = off[expert_ind.long()] + start_ind.long() - self.v_margin
        rows = base.unsqueeze(0) + torch.arange(self.v_margin + length, device=table.device).unsqueeze(1)
        return table[rows]

    def context(self, window):
        """(v_margin, B, D) -> the per-episode video context (B, v_hdim): last output of the causal net."""
        return self.forward_v_net(window[:self.v_margin])[-1]

    def s_step(self, state, hc):
        """One step of the state net for a batch of envs: state (B, state_dim), hc = (h, c) or None -> (out, (h, c))."""
        if self.s_net_type != "lstm":
            return state, hc
        h, c = self.s_net.rnn_f(state, hc)
        return h, (h, c)

    def initialize(self, x):
        if self.mode == "test":
            p = next(self.parameters())
            x = x.to(device=p.device, dtype=p.dtype)
            self.v_out = self.context(x.unsqueeze(1) if x.dim() == 2 else x)
            if self.s_net_type == "lstm":
                self.s_net.initialize(self.v_out.shape[0])
            self.t = 0
            return
        masks, cnn_feat, v_metas = x
        device, dtype = masks.device, masks.dtype
        m = self.v_margin
        ends = torch.nonzero(masks == 0).flatten().cpu().numpy()
        n = masks.shape[0]
        starts = np.concatenate(([0], ends[:-1] + 1))
        lens = ends - starts + 1
        max_len = _dist.global_max(int(lens.max()), device)       # batch-wide in the reference; global when sharded
        self.num_episode, self.max_episode_len = len(ends), max_len
        idx = np.arange(n)
        covered = int(ends[-1]) + 1
        idx[:covered] = np.repeat(np.arange(len(ends)), lens) * max_len + (np.arange(covered) - np.repeat(starts, lens))
        self.indices = idx
        meta = np.asarray(v_metas)[ends]
        T_ctx = m + max_len if self.dynamic_v else m
        if self._cnn_table is not None and self._cnn_table[0].device == device:
            # episode count rounded up to a bucket (gemm_tuning.py); nothing gathers from the extra columns
            pad = _tuning.pad_to(len(ends), _tuning.EPISODE_BUCKET) if _tuning.enabled() and len(ends) >= 4 * _tuning.EPISODE_BUCKET else 0
            if pad:
                meta = np.concatenate((meta, np.repeat(meta[:1], pad, axis=0)), 0)
                self.num_episode = meta.shape[0]
            self.check_windows(meta[:, 0], meta[:, 1])
            win = self.window_features(torch.as_tensor(meta[:, 0], device=device), torch.as_tensor(meta[:, 1], device=device)).to(dtype)
            ctx = win.new_zeros(T_ctx, meta.shape[0], self.cnn_feat_dim)
            ctx[:m] = win
        else:
            ctx = np.zeros((T_ctx, len(ends), self.cnn_feat_dim))
            for e, (ei, si) in enumerate(meta):
                ctx[:m, e, :] = cnn_feat[int(ei)][int(si) - m: int(si)]
            ctx = torch.as_tensor(ctx, dtype=dtype, device=device)
        self.cnn_feat_ctx = ctx
        self.gather_indices = torch.as_tensor(idx, dtype=torch.long, device=device)
        self._ctx_key = (int(max_len), meta.shape[0], hash(meta.tobytes()))       # which windows / episodes this batch holds
        self._grp = None

    def state_sequences(self, x):
        """Flat batch of states -> (max_episode_len, num_episode, state_dim), zero where an episode is shorter."""
        s_ctx = x.new_zeros(self.num_episode * self.max_episode_len, self.state_dim)
        s_ctx = s_ctx.index_copy(0, self.gather_indices, x)
        return s_ctx.view(self.num_episode, self.max_episode_len, self.state_dim).transpose(0, 1).contiguous()

    def forward(self, x):
        if self.mode == "test":
            if self.s_net_type == "lstm":
                x = self.s_net(x)
            self.t += 1
            return torch.cat((self.v_out, x), dim=1)
        v_seq = s_seq = None
        if self._grp is not None:            # both sweeps came out of grouped launches (grouped_forecast_context)
            (v_seq, s_seq, x_ref, with_grad), self._grp = self._grp, None
            if x_ref is not x or with_grad != torch.is_grad_enabled():
                v_seq = s_seq = None         # prepared for another batch / autograd mode: never reuse
        if v_seq is None:
            v_seq = self.forward_v_net(self.cnn_feat_ctx)
        v_ctx = v_seq[self.v_margin:] if self.dynamic_v else v_seq[[-1]].expand(self.max_episode_len, -1, -1)
        v_out = v_ctx.transpose(0, 1).reshape(-1, self.v_hdim).index_select(0, self.gather_indices)
        if self.s_net_type == "lstm":
            if s_seq is None:
                s_seq = self.s_net(self.state_sequences(x))
            s_out = s_seq.transpose(0, 1).reshape(-1, self.s_hdim).index_select(0, self.gather_indices)
        else:
            s_out = x
        return torch.cat((v_out, s_out), dim=1)


def grouped_video_context(nets):
    """Train-mode video contexts of several VideoStateNets (the critic's and the actor's) in ONE grouped recurrent
    launch each way (lstm.LstmGroup): their bi-LSTMs read the same windows with different weights, and a single sweep
    leaves most of the chip idle. Each net's next forward() consumes its context. Returns False (nothing done) when the
    nets do not qualify; the nets then compute their contexts one by one as usual."""
    if _LSTM_IMPL == "torch" or len(nets) < 1:
        return False
    n0 = nets[0]
    for n in nets:
        if not (isinstance(n, VideoStateNet) and n.mode == "train" and n._buckets is None and n.v_net.cell_type == "lstm"
                and n.cnn_feat_ctx is not None and n.cnn_feat_ctx.shape == n0.cnn_feat_ctx.shape and n._ctx_key == n0._ctx_key
                and n.cnn_feat_ctx.dtype == n0.cnn_feat_ctx.dtype and n.v_net.bi_dir == n0.v_net.bi_dir):
            return False
    cells, revs = [], []
    for n in nets:
        cells.append(n.v_net.rnn_f); revs.append(False)
        if n.v_net.bi_dir:
            cells.append(n.v_net.rnn_b); revs.append(True)
    x = n0.cnn_feat_ctx                      # same episodes, same windows for every net (initialize() of the same batch)
    if not _hip_lstm.group_available(x, cells):
        return False
    hs = _hip_lstm.lstm_group(x, cells, revs, pairs=n0.v_net.bi_dir, ragged=n0._ragged,     # bi-directional: (T, B, 2H) per net, no concatenation
                              frames=getattr(n0, "_frames", None))
    for i, n in enumerate(nets):
        n._v_ctx = (hs[i], torch.is_grad_enabled())
    return True


def grouped_forecast_context(nets, x):
    """ego_forecast's counterpart of grouped_video_context: the causal video LSTMs of all nets in one grouped launch each
    way, and their state LSTMs over the scattered states `x` in another. Each net's next forward(x) consumes its pair."""
    if _LSTM_IMPL == "torch" or len(nets) < 2 or x is None:
        return False
    n0 = nets[0]
    for n in nets:
        if not (isinstance(n, VideoForecastNet) and n.mode == "train" and n.s_net_type == "lstm" and n.cnn_feat_ctx is not None
                and n._ctx_key is not None and n._ctx_key == n0._ctx_key and n.cnn_feat_ctx.shape == n0.cnn_feat_ctx.shape
                and n.cnn_feat_ctx.dtype == n0.cnn_feat_ctx.dtype and n.dynamic_v == n0.dynamic_v and n.state_dim == n0.state_dim
                and not n.v_net.bi_dir and not n.s_net.bi_dir and n.v_net.cell_type == "lstm" and n.s_net.cell_type == "lstm"):
            return False
    v_cells, s_cells = [n.v_net.rnn_f for n in nets], [n.s_net.rnn_f for n in nets]
    s_in = n0.state_sequences(x)
    if not (_hip_lstm.group_available(n0.cnn_feat_ctx, v_cells) and _hip_lstm.group_available(s_in, s_cells)):
        return False
    fwd = [False] * len(nets)
    hv = _hip_lstm.lstm_group(n0.cnn_feat_ctx, v_cells, fwd)
    hs = _hip_lstm.lstm_group(s_in, s_cells, fwd)
    for n, a, b in zip(nets, hv, hs):
        n._grp = (a, b, x, torch.is_grad_enabled())
    return True


# ---------------------------------------------------------------------- image encoder of the state regressor
class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = torch.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return torch.relu(out + idt)


class ResNet18(nn.Module):
    """The published ResNet-18 (He et al. 2016; basic blocks [2, 2, 2, 2], 64-128-256-512 channels) with the module and
    parameter names of torchvision's `resnet18`, so a torchvision state dict loads unchanged; `fc` -> out_dim as
    models/resnet.py:12-17 does. torchvision is not in this image, so no pretrained weights are fetched here."""

    def __init__(self, out_dim):
        super().__init__()
        self.out_dim = out_dim
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        chans, layers, cin = (64, 128, 256, 512), [], 64
        for i, c in enumerate(chans):
            layers.append(nn.Sequential(_BasicBlock(cin, c, 1 if i == 0 else 2), _BasicBlock(c, c, 1)))
            cin = c
        self.layer1, self.layer2, self.layer3, self.layer4 = layers
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, out_dim)

    def forward(self, x):
        x = self.maxpool(torch.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class ResNet(nn.Module):
    """models/resnet.py:6-18: `self.resnet` = ResNet-18 whose fc maps to `out_dim`."""

    def __init__(self, out_dim, fix_params=False):
        super().__init__()
        self.out_dim = out_dim
        self.resnet = ResNet18(out_dim)
        if fix_params:
            for name, p in self.resnet.named_parameters():
                p.requires_grad = name.startswith("fc.")

    def forward(self, x):
        return self.resnet(x)
