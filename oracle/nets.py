"""ORACLE (test infrastructure): policy / value / video-context nets as pure functions.

Functional torch-CPU float64 restatement (parameters passed as dicts keyed like the
reference ``state_dict``) of:
  MLP.forward                       models/mlp.py:22-25
  PolicyGaussian.forward/log_prob   core/policy_gaussian.py:19-24, core/distributions.py:21-22
  Value.forward                     core/critic.py:15-18
  RNN.batch_forward (bi-LSTM of LSTMCells)   models/rnn.py:45-61
  VideoStateNet test / train modes  models/video_state_net.py:36-70
Pinned against tests/golden/{policy_value,video_state_net,ppo_update}.npz.
"""
import math

import numpy as np
import torch

F64 = torch.float64


def as_t(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x), dtype=F64)


def params_from_npz(npz, prefix):
    return {k[len(prefix):]: as_t(npz[k]).clone() for k in npz.files if k.startswith(prefix)}


def mlp(p, x, prefix="net.", act=torch.relu):
    i = 0
    while "%saffine_layers.%d.weight" % (prefix, i) in p:
        x = act(x @ p["%saffine_layers.%d.weight" % (prefix, i)].T + p["%saffine_layers.%d.bias" % (prefix, i)])
        i += 1
    return x


def policy_mean_std(p, x, act=torch.relu):
    h = mlp(p, x, act=act)
    mean = h @ p["action_mean.weight"].T + p["action_mean.bias"]
    std = torch.exp(p["action_log_std"].expand_as(mean))
    return mean, std


def gaussian_log_prob(mean, std, a):
    var = std * std
    lp = -((a - mean) ** 2) / (2 * var) - torch.log(std) - 0.5 * math.log(2 * math.pi)
    return lp.sum(1, keepdim=True)


def value(p, x, act=torch.relu):
    h = mlp(p, x, act=act)
    return h @ p["value_head.weight"].T + p["value_head.bias"]


def lstm_cell(p, pre, x, h, c):
    g = x @ p[pre + "weight_ih"].T + p[pre + "bias_ih"] + h @ p[pre + "weight_hh"].T + p[pre + "bias_hh"]
    i, f, gg, o = g.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def bilstm(p, x, prefix="v_net."):
    """x (T,B,D) -> (T,B,2H): forward cell over t ascending, backward cell over t descending."""
    T, B = x.shape[0], x.shape[1]
    H = p[prefix + "rnn_f.weight_hh"].shape[1]
    outs = []
    for pre, order in ((prefix + "rnn_f.", range(T)), (prefix + "rnn_b.", range(T - 1, -1, -1))):
        h = torch.zeros(B, H, dtype=x.dtype)
        c = torch.zeros(B, H, dtype=x.dtype)
        seq = [None] * T
        for t in order:
            h, c = lstm_cell(p, pre, x[t], h, c)
            seq[t] = h
        outs.append(torch.stack(seq, 0))
    return torch.cat(outs, 2)


def vsnet_test_init(p, window, margin):
    """window (T+2m, D) -> v_out (T, 2H)."""
    out = bilstm(p, as_t(window).unsqueeze(1)).squeeze(1)
    return out[margin:-margin]


def episode_layout(masks, v_metas):
    """Episode segmentation of a flat batch (video_state_net.py:41-52).

    returns (indices (N,), ep_meta (n_ep,2), max_len) with indices[i] = ep*max_len + t.
    """
    masks = np.asarray(masks).ravel()
    ends = np.where(masks == 0)[0]
    starts = np.concatenate([[0], ends[:-1] + 1])
    lens = ends - starts + 1
    max_len = int(lens.max())
    idx = np.arange(masks.shape[0])
    for e, (s, en) in enumerate(zip(starts, ends)):
        idx[s:en + 1] = e * max_len + np.arange(en - s + 1)
    # samples after the last mask==0 (none in a well-formed batch) keep their raw index, as the reference
    return idx, np.asarray(v_metas)[ends], max_len


def vsnet_train_ctx(masks, cnn_feat, v_metas, margin, cdim):
    idx, ep_meta, max_len = episode_layout(masks, v_metas)
    ctx = np.zeros((max_len + 2 * margin, len(ep_meta), cdim))
    for e, (ei, si) in enumerate(ep_meta):
        ctx[:, e] = cnn_feat[int(ei)][int(si) - margin: int(si) + max_len + margin]
    return idx, ctx


def vsnet_train_forward(p, ctx, idx, states, margin):
    v = bilstm(p, as_t(ctx))[margin:-margin]
    v = v.transpose(0, 1).contiguous().view(-1, v.shape[-1])
    return torch.cat([v[torch.as_tensor(idx, dtype=torch.long)], as_t(states)], dim=1)


def resnet18_forward(sd, x, prefix="", train=False, eps=1e-5):
    """ResNet-18 (He et al. 2016, the layout of torchvision's `resnet18`, which models/resnet.py:6-18 wraps with `fc` ->
    out_dim) written with torch.nn.functional calls over a state dict with torchvision's key names -- an independent
    restatement of the published architecture (torchvision itself is not in the image): stem conv 7x7/2 pad 3 -> BN -> ReLU ->
    max-pool 3x3/2 pad 1; layers 1-4 = two basic blocks each (conv 3x3 -> BN -> ReLU -> conv 3x3 -> BN, + identity or a
    1x1/2 conv + BN projection, ReLU), the first block of layers 2-4 strides by 2; global average pool; fc.
    `train`: batch statistics in the normalisation layers (running statistics are not updated here)."""
    import torch.nn.functional as F
    g = lambda k: sd[prefix + k]

    def bn(h, name):
        return F.batch_norm(h, None if train else g(name + ".running_mean"), None if train else g(name + ".running_var"),
                            g(name + ".weight"), g(name + ".bias"), training=train, momentum=0.0, eps=eps)

    h = F.relu(bn(F.conv2d(x, g("conv1.weight"), None, stride=2, padding=3), "bn1"))
    h = F.max_pool2d(h, kernel_size=3, stride=2, padding=1)
    for li in range(1, 5):
        for b in range(2):
            p = "layer%d.%d" % (li, b)
            stride = 2 if (b == 0 and li > 1) else 1
            o = F.relu(bn(F.conv2d(h, g(p + ".conv1.weight"), None, stride=stride, padding=1), p + ".bn1"))
            o = bn(F.conv2d(o, g(p + ".conv2.weight"), None, stride=1, padding=1), p + ".bn2")
            if (prefix + p + ".downsample.0.weight") in sd:
                h = bn(F.conv2d(h, g(p + ".downsample.0.weight"), None, stride=stride, padding=0), p + ".downsample.1")
            h = F.relu(o + h)
    h = F.adaptive_avg_pool2d(h, 1).flatten(1)
    return F.linear(h, g("fc.weight"), g("fc.bias"))
