"""float32 matrix products of the PPO update through `egp_gemm_f32` (csrc/egp_gemm.hip): bf16 matrix cores with split
operands (every float32 value as a sum of bf16 pieces, float32-class products), bias / ReLU / dReLU fused into
the epilogues, deterministic split-K weight gradients that also return the bias gradient.

  linear_fwd(x, W, b, relu)          y = x W^T + b        (nn.Linear of models/mlp.py:22-25, core/policy_gaussian.py:19-24)
  linear_dgrad(dy, W, mask)          dx = (dy W) * (mask > 0)
  linear_wgrad(dy, x)                dW = dy^T x, db = sum_rows dy
  mlp_head(x, hidden_layers, head)   the reference's `head(MLP(x))` as ONE autograd node: activations saved once, every
                                     ReLU derivative applied inside the producing data-gradient product

`EGP_GEMM=torch` keeps every product on the library path (rocBLAS / hipBLASLt through torch); `EGP_GEMM_TERMS` picks the
operand split: 6 (default: three bf16 pieces, float32-class products), 3 (two pieces, ~16 mantissa bits) or 1 (plain bf16).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L

_WS = {}


def enabled():
    return os.environ.get("EGP_GEMM", "hip") != "torch"


def default_terms():
    """Pieces per operand / MFMAs per product: 6 = three bf16 pieces, every cross term down to 2^-16 (float32-class
    products, the default); 3 = two pieces (~16 mantissa bits); 1 = plain bf16 inputs."""
    t = os.environ.get("EGP_GEMM_TERMS", "6")
    if t not in ("1", "3", "6"):
        raise ValueError("EGP_GEMM_TERMS must be 1, 3 or 6, got %r" % t)
    return int(t)


_WS_MAX = 8


def _workspace(n_floats, device):
    """Split-K / bias-gradient scratch, one per (device, stream): products issued on two streams must not share it. Bounded,
    least recently used first out (as optim._zeroed_workspace): callers with short-lived streams must not grow it for ever."""
    key = (str(device), int(torch.cuda.current_stream(device).cuda_stream))
    buf = _WS.pop(key, None)
    if buf is None or buf.numel() < n_floats:
        buf = torch.empty(max(int(n_floats), 1 << 20), dtype=torch.float32, device=device)
    _WS[key] = buf                     # (re-inserted: dict order = recency)
    while len(_WS) > _WS_MAX:
        _WS.pop(next(iter(_WS)))
    return buf


def _mat(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2):
        raise ValueError("%s must be a 2-D float32 HIP tensor, got %s %s" % (name, tuple(t.shape), t.dtype))
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise ValueError("%s must have unit stride along its second dimension (strides %s)" % (name, t.stride()))
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


_CUS = {}


def usable_cus(device=None):
    """CUs this process really gets on `device` (egp_device_usable_cus: 256 on a whole MI355X, fewer under a CU mask)."""
    if not torch.cuda.is_available():
        return 256
    dev = torch.cuda.current_device() if device is None else int(device)
    if dev not in _CUS:
        n = int(L.load().egp_device_usable_cus(dev))
        _CUS[dev] = n if n > 0 else 256
    return _CUS[dev]


def pick_splits(M, n_out, K):
    """Split-K factor of a weight-gradient shaped product: one work item per CU of the persistent kernel (tiles x splits <=
    256, rounded DOWN: 6 tiles x 86 splits = 516 items used to mean a third round for four workgroups), at least 8 k-tiles
    each. Measured on the update's shapes against two items per CU: the products themselves take the same time, the
    split-K reductions half (1.29 -> 0.74 ms per update)."""
    cus = usable_cus()
    if M == 1:          # a one-column dy (the value head): k_colsum streams the activations, one workgroup per split -- one per CU
        return max(1, min(cus, K // 128))      # (128 / 256 / 512 / 1 024 splits: 52.9 / 37.8 / 41.3 / 61.2 us for 134 k x 200)
    tiles = ((M + 127) // 128) * ((n_out + 127) // 128)
    return max(1, min(cus // tiles, K // 256))


def _index(t, name):
    if not (t.is_cuda and t.dtype == torch.int64 and t.dim() == 1 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous 1-D int64 HIP tensor" % name)
    return t


def gemm(A, B, a_kcontig=True, b_kcontig=True, bias=None, relu=False, mask=None, terms=None, splits=1, want_bias_grad=False,
         out=None, bias_grad_out=None, accumulate=False, a_rows=None, a2=None, b_krows=None, b2=None, c_rows=None, a_krows=None):
    """C = A B with the operand forms of include/egopose_hip.h (`egp_gemm_desc`). A: (M, K) if a_kcontig else (K, M);
    B: (N, K) if b_kcontig else (K, N). Returns C, or (C, bias_grad) with want_bias_grad.
    Fused gather / scatter (three-piece products only): `a_rows` (int64, M entries) -- operand row m is A[a_rows[m]], and with
    `a2` (M, K2) its columns continue with a2[m] (K = A.shape[1] + K2); `b_krows` / `b2` the same for B given as (K, N) along
    k and n; `c_rows` -- result row m goes to out[c_rows[m]] (`out` required, rows nobody writes keep their content);
    `a_krows` -- A given as (K, M): k-row k is A[a_krows[k]] (with `b_krows`: a weight gradient over a subset of the rows)."""
    A, B = _mat(A, "A"), _mat(B, "B")
    M, K = (A.shape if a_kcontig else A.shape[::-1])
    N, Kb = (B.shape if b_kcontig else B.shape[::-1])
    a_split = b_split = 0
    if a_rows is not None or a2 is not None:
        if not a_kcontig:
            raise ValueError("a_rows / a2 go with a k-contiguous A")
        if a_rows is not None:
            M = _index(a_rows, "a_rows").shape[0]
        if a2 is not None:
            _mat(a2, "a2")
            if a2.shape[0] != M:
                raise ValueError("a2 has %d rows, the operand %d" % (a2.shape[0], M))
            a_split, K = K, K + a2.shape[1]
    if a_krows is not None:
        if a_kcontig:
            raise ValueError("a_krows goes with A given as (K, M)")
        K = _index(a_krows, "a_krows").shape[0]
    if b_krows is not None or b2 is not None:
        if b_kcontig:
            raise ValueError("b_krows / b2 go with B given as (K, N)")
        if b_krows is not None:
            Kb = _index(b_krows, "b_krows").shape[0]
        if b2 is not None:
            _mat(b2, "b2")
            if b2.shape[0] != Kb:
                raise ValueError("b2 has %d rows, the operand %d" % (b2.shape[0], Kb))
            b_split, N = N, N + b2.shape[1]
    if K != Kb:
        raise ValueError("inner dimensions differ: %d vs %d" % (K, Kb))
    dev = A.device
    if c_rows is not None:
        if out is None or _index(c_rows, "c_rows").shape[0] != M:
            raise ValueError("c_rows needs `out` (the scatter destination) and one entry per result row")
    elif out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=dev)
    _mat(out, "out")
    d = L.GemmDesc()
    d.M, d.N, d.K = M, N, K
    if a_rows is not None:
        d.a_rows, d.a_src_rows = a_rows.data_ptr(), A.shape[0]
    if a2 is not None:
        d.A2, d.lda2, d.a_split = a2.data_ptr(), _ld(a2), a_split
    if b_krows is not None:
        d.b_krows, d.b_src_rows = b_krows.data_ptr(), B.shape[0]
    if b2 is not None:
        d.B2, d.ldb2, d.b_split = b2.data_ptr(), _ld(b2), b_split
    if c_rows is not None:
        d.c_rows = c_rows.data_ptr()
    if a_krows is not None:
        d.a_krows = a_krows.data_ptr()
    d.A, d.lda, d.a_kcontig = A.data_ptr(), _ld(A), 1 if a_kcontig else 0
    d.B, d.ldb, d.b_kcontig = B.data_ptr(), _ld(B), 1 if b_kcontig else 0
    d.C, d.ldc = out.data_ptr(), _ld(out)
    d.bias = bias.data_ptr() if bias is not None else None
    d.relu = 1 if relu else 0
    if mask is not None:
        _mat(mask, "mask")
        d.mask, d.ldmask = mask.data_ptr(), _ld(mask)
    d.terms = int(terms or default_terms())
    d.splits = int(splits)
    d.accumulate = 1 if accumulate else 0
    bg = None
    if want_bias_grad:
        bg = bias_grad_out if bias_grad_out is not None else torch.empty(M, dtype=torch.float32, device=dev)
        d.bias_grad = bg.data_ptr()
    if splits > 1 or want_bias_grad:
        ws = _workspace(L.load().egp_gemm_workspace_floats(M, N, 1 if want_bias_grad else 0, int(splits)), dev)
        d.workspace = ws.data_ptr()
    L.check(L.load().egp_gemm_f32(C.byref(d), L.current_stream()), "egp_gemm_f32")
    return (out, bg) if want_bias_grad else out


def linear_fwd(x, W, b=None, relu=False, terms=None):
    return gemm(x, W, True, True, bias=b, relu=relu, terms=terms)


def linear_dgrad(dy, W, mask=None, n_cols=None, terms=None):
    """dx[:, :n_cols] = (dy W[:, :n_cols]) * (mask > 0)."""
    Wv = W if n_cols is None else W[:, :n_cols]
    return gemm(dy, Wv, True, False, mask=mask, terms=terms)


def linear_wgrad(dy, x, want_bias=True, terms=None):
    """dW (out, in) = dy^T x [and db (out,) = column sums of dy], split-K over the batch rows.
    Returns (dW, db) with want_bias, else dW."""
    K, M = dy.shape
    n_out = x.shape[1] + (1 if want_bias else 0)
    return gemm(dy, x, False, False, terms=terms, splits=pick_splits(M, n_out, K), want_bias_grad=want_bias)


class MlpHead(torch.autograd.Function):
    """out = head(relu-MLP(x)): apply(x, n_grad_cols, W1, b1, ..., Wh, bh). Gradient w.r.t. x only for its first
    n_grad_cols columns (the video context; the state columns of the reference's concatenated input need none),
    zeros elsewhere."""

    @staticmethod
    def forward(ctx, x, n_grad_cols, *params):
        Ws, bs = params[0::2], params[1::2]
        hs = [x.contiguous()]
        for i, (W, b) in enumerate(zip(Ws, bs)):
            hs.append(linear_fwd(hs[-1], W.contiguous(), b, relu=i < len(Ws) - 1))
        ctx.save_for_backward(*hs[:-1], *Ws)
        ctx.n_layers, ctx.n_grad_cols = len(Ws), int(n_grad_cols)
        return hs[-1]

    @staticmethod
    def backward(ctx, dout):
        nl = ctx.n_layers
        hs, Ws = ctx.saved_tensors[:nl], ctx.saved_tensors[nl:]
        dz = dout.contiguous()
        grads = [None] * (2 * nl)
        dx = None
        for i in range(nl - 1, -1, -1):
            need_w, need_b = ctx.needs_input_grad[2 + 2 * i], ctx.needs_input_grad[3 + 2 * i]
            if need_w or need_b:
                dW, db = linear_wgrad(dz, hs[i], want_bias=True)
                grads[2 * i], grads[2 * i + 1] = (dW if need_w else None), (db if need_b else None)
            if i > 0:
                dz = linear_dgrad(dz, Ws[i].contiguous(), mask=hs[i])         # hs[i] = relu output feeding layer i
            elif ctx.needs_input_grad[0]:
                nc = ctx.n_grad_cols
                W0 = Ws[0].contiguous()
                if nc >= W0.shape[1]:
                    dx = linear_dgrad(dz, W0)
                else:
                    dx = torch.zeros(dz.shape[0], W0.shape[1], dtype=dz.dtype, device=dz.device)
                    gemm(dz, W0[:, :nc], True, False, out=dx[:, :nc])
        return (dx, None, *grads)


def mlp_head_available(x, layers, head, activation):
    if isinstance(x, GatheredInput):
        x = x.x
    if not (enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and activation is torch.relu):
        return False
    return all(l.weight.dtype == torch.float32 and l.bias is not None and l.weight.is_cuda for l in list(layers) + [head])


def mlp_head(x, layers, head, n_grad_cols=None):
    params = []
    for l in list(layers) + [head]:
        params += [l.weight, l.bias]
    nc = x.shape[1] if n_grad_cols is None else int(n_grad_cols)
    return MlpHead.apply(x, nc, *params)


class GatheredInput:
    """[ctx2d[idx] | x] that has not been formed: what VideoStateNet.forward('train') hands to a policy / value MLP that can
    read the two tensors itself (GatherMlpHead). Anything else calls `materialize()` (GatherConcat)."""

    def __init__(self, ctx2d, idx, x):
        self.ctx2d, self.idx, self.x = ctx2d, idx, x
        self.shape = (x.shape[0], ctx2d.shape[1] + x.shape[1])
        self.dtype, self.device, self.is_cuda = x.dtype, x.device, x.is_cuda

    def dim(self):
        return 2

    def size(self, k=None):
        return torch.Size(self.shape) if k is None else self.shape[k]

    def materialize(self):
        return GatherConcat.apply(self.ctx2d, self.idx, self.x)

    def __getitem__(self, ind):
        if torch.is_tensor(ind) and ind.dtype == torch.int64 and ind.dim() == 1:      # a row subset stays lazy (rows stay unique)
            return GatheredInput(self.ctx2d, self.idx.index_select(0, ind), self.x.index_select(0, ind))
        return self.materialize()[ind]


def fused_gather_available(H, n_hidden, S):
    """The first MLP layer can read [ctx[idx] | state] itself: three-piece products on the persistent kernel, the context
    width H a multiple of the 128-column tile (the weight gradient switches source between column tiles), at least one
    k-tile of state columns S (a k-tile reads one source)."""
    import os
    return enabled() and default_terms() == 6 and os.environ.get("EGP_GEMM_WS", "1") != "0" and H % 128 == 0 and n_hidden % 4 == 0 \
        and S >= 32


def fused_rows_available():
    """Row / k-row index operands (a_rows, c_rows, a_krows, b_krows) need three-piece products on the persistent kernel."""
    import os
    return enabled() and default_terms() == 6 and os.environ.get("EGP_GEMM_WS", "1") != "0"


class GatherMlpHead(torch.autograd.Function):
    """out = head(relu-MLP([ctx2d[idx] | x])) without forming the concatenated input: apply(ctx2d, idx, x, W1, b1, ..., Wh, bh).
    The first layer's product gathers the context rows and appends the state columns on its way into LDS
    (`egp_gemm_desc.a_rows / A2`), its weight gradient does the same along k (`b_krows / B2`), and its data gradient writes the
    context rows it belongs to (`c_rows`; idx must not repeat). Gradient w.r.t. ctx2d only, as GatherConcat + MlpHead."""

    @staticmethod
    def forward(ctx, ctx2d, idx, x, *params):
        Ws, bs = params[0::2], params[1::2]
        ctx2d, x = ctx2d.contiguous(), (x if x.stride(1) == 1 else x.contiguous())
        h = gemm(ctx2d, Ws[0].contiguous(), True, True, bias=bs[0], relu=len(Ws) > 1, a_rows=idx, a2=x)
        hs = [h]
        for i in range(1, len(Ws)):
            hs.append(linear_fwd(hs[-1], Ws[i].contiguous(), bs[i], relu=i < len(Ws) - 1))
        ctx.save_for_backward(ctx2d, idx, x, *hs[:-1], *Ws)
        ctx.n_layers = len(Ws)
        return hs[-1]

    @staticmethod
    def backward(ctx, dout):
        nl = ctx.n_layers
        ctx2d, idx, x = ctx.saved_tensors[:3]
        hs, Ws = ctx.saved_tensors[3:3 + nl - 1], ctx.saved_tensors[3 + nl - 1:]        # hs[i - 1] = relu output feeding layer i
        dz = dout.contiguous()
        grads = [None] * (2 * nl)
        for i in range(nl - 1, 0, -1):
            need_w, need_b = ctx.needs_input_grad[3 + 2 * i], ctx.needs_input_grad[4 + 2 * i]
            if need_w or need_b:
                dW, db = linear_wgrad(dz, hs[i - 1], want_bias=True)
                grads[2 * i], grads[2 * i + 1] = (dW if need_w else None), (db if need_b else None)
            dz = linear_dgrad(dz, Ws[i].contiguous(), mask=hs[i - 1])
        H, S = ctx2d.shape[1], x.shape[1]
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            n = dz.shape[0]
            dW, db = gemm(dz, ctx2d, False, False, splits=pick_splits(dz.shape[1], H + S + 1, n), want_bias_grad=True, b_krows=idx, b2=x)
            grads[0], grads[1] = (dW if ctx.needs_input_grad[3] else None), (db if ctx.needs_input_grad[4] else None)
        dctx = None
        if ctx.needs_input_grad[0]:
            dctx = torch.zeros_like(ctx2d)
            gemm(dz, Ws[0][:, :H], True, False, out=dctx, c_rows=idx)
        return (dctx, None, None, *grads)


def gather_mlp_head(gi, layers, head):
    """head(relu-MLP([ctx2d[idx] | x])), one launch per layer and direction (GatherMlpHead)."""
    params = []
    for l in list(layers) + [head]:
        params += [l.weight, l.bias]
    return GatherMlpHead.apply(gi.ctx2d, gi.idx, gi.x, *params)


class GatherConcat(torch.autograd.Function):
    """out[i] = [ctx2d[idx[i]] | x[i]] (`egp_gather_concat_f32`); gradient to ctx2d only (idx must not repeat)."""

    @staticmethod
    def forward(ctx, ctx2d, idx, x):
        n, H, S = idx.shape[0], ctx2d.shape[1], x.shape[1]
        out = torch.empty(n, H + S, dtype=torch.float32, device=x.device)
        L.check(L.load().egp_gather_concat_f32(ctx2d.data_ptr(), _ld(ctx2d), idx.data_ptr(), x.data_ptr(), _ld(x), n, H, S,
                                               out.data_ptr(), H + S, L.current_stream()), "egp_gather_concat_f32")
        ctx.save_for_backward(idx)
        ctx.shape = tuple(ctx2d.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None
        R, H = ctx.shape
        dout = _mat(dout if dout.stride(1) == 1 else dout.contiguous(), "dout")
        dctx = torch.zeros(R, H, dtype=torch.float32, device=dout.device)
        L.check(L.load().egp_scatter_rows_f32(dout.data_ptr(), _ld(dout), idx.data_ptr(), idx.shape[0], H, dctx.data_ptr(), H,
                                              L.current_stream()), "egp_scatter_rows_f32")
        return dctx, None, None


def gather_concat_available(ctx2d, idx, x):
    return (enabled() and x.is_cuda and ctx2d.dtype == torch.float32 and x.dtype == torch.float32 and ctx2d.dim() == 2 and x.dim() == 2
            and ctx2d.is_contiguous() and x.stride(1) == 1 and idx.dtype == torch.int64 and idx.is_contiguous() and not x.requires_grad)
