"""The update's policy / value MLP as one HIP launch per direction (csrc/egp_chain.hip: `egp_mlp_chain_f32`) -- OPT-IN.

`head(relu-MLP([ctx2d[idx] | x]))` of the reference (models/video_state_net.py:65-69 -> models/mlp.py:22-25 ->
core/policy_gaussian.py:19-24 / core/critic.py:15-18) for the whole batch: the forward launch walks the three layers per
128-row tile with the activations in registers and saves x^T, h1^T, h2^T ([feature][row]); the backward launch walks the
data-gradient chain the same way and leaves d z^T for the three weight gradients, which are k-contiguous products of
`gemm.gemm` (both operands [feature][row]). Built for the shipped widths (hidden 300 / 200, context 128, output 52 | 1).

Status (round 4, measured on the MI355X at the bench's 134 656 rows, tools/probes/chain_time.py): results agree with a float64
evaluation to 3e-6 (tests/test_chain_gpu.py), but the launches are SLOWER than the layer-per-launch path they were meant to
replace -- forward 0.58 against 0.38 ms, forward + backward 1.57 against 1.23 ms per net -- so the default stays
`gemm.GatherMlpHead`; `EGP_MLP_CHAIN=1` switches the chain on. Why, by ablation (tools/probes/chain_exp.sh, phase stamps of one
tile): a 128-row tile takes 108 us where its 1 914 MFMAs alone take ~36 us (19 ns per v_mfma_f32_32x32x16_bf16 at the clock the
chip sustains); 48 stage boundaries cost ~0.32 us each (barrier, B-fragment conversion, first LDS round trip), the epilogues'
[feature][row] stores 18 us, the input gather and the LDS-DMA weight stream another ~28 us of stalls. The register chaining
needs 440 VGPRs, i.e. ONE wave per SIMD, so nothing covers those stalls -- the layer-per-launch kernels run two waves per SIMD
at 48 % matrix-pipe utilisation against this kernel's 26 %. What would have to change is in DESIGN.md (section 0, item 1).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from . import gemm as G

_HIDDEN = (300, 200)
_CTX = 128
_OUTS = (52, 1)


def enabled():
    return G.enabled() and G.default_terms() == 6 and os.environ.get("EGP_MLP_CHAIN", "0") == "1"


def available(gi, layers, head):
    """`gi`: gemm.GatheredInput; layers: the MLP's affine layers; head: the output Linear."""
    layers = list(layers)
    if not (enabled() and isinstance(gi, G.GatheredInput) and len(layers) == 2):
        return False
    if tuple(l.out_features for l in layers) != _HIDDEN or head.out_features not in _OUTS:
        return False
    H, S = gi.ctx2d.shape[1], gi.x.shape[1]
    if H != _CTX or layers[0].in_features != H + S or H + S > 16 * 20:
        return False
    ok = lambda t: t.is_cuda and t.dtype == torch.float32
    return (ok(gi.ctx2d) and ok(gi.x) and gi.ctx2d.is_contiguous() and gi.x.stride(1) == 1 and gi.idx.dtype == torch.int64
            and gi.idx.is_contiguous() and all(ok(l.weight) and l.bias is not None for l in layers + [head]))


def _pack(W, n_rows, n_k, transpose, chained):
    lib = L.load()
    nbytes = int(lib.egp_mlp_chain_pack_bytes(n_rows, n_k))
    buf = torch.empty(nbytes, dtype=torch.uint8, device=W.device)
    W = W if W.stride(-1) == 1 else W.contiguous()
    L.check(lib.egp_mlp_chain_pack_f32(C.c_void_p(W.data_ptr()), int(W.stride(0)), n_rows, n_k, 1 if transpose else 0, 1 if chained else 0,
                                       C.c_void_p(buf.data_ptr()), L.current_stream()), "egp_mlp_chain_pack_f32")
    return buf


def _pad32(b):
    n = (b.shape[0] + 31) // 32 * 32
    out = torch.zeros(n, dtype=torch.float32, device=b.device)
    out[:b.shape[0]] = b
    return out


def _p(t):
    return t.data_ptr() if t is not None else None


class ChainMlpHead(torch.autograd.Function):
    """out = head(relu-MLP([ctx2d[idx] | x])): apply(ctx2d, idx, x, W1, b1, W2, b2, W3, b3). Gradients: ctx2d (rows idx, which
    must not repeat; other rows zero) and the six parameters -- as gemm.GatherMlpHead."""

    @staticmethod
    def forward(ctx, ctx2d, idx, x, W1, b1, W2, b2, W3, b3):
        lib = L.load()
        n, H, S = idx.shape[0], ctx2d.shape[1], x.shape[1]
        N1, N2, N3 = W1.shape[0], W2.shape[0], W3.shape[0]
        dev = x.device
        need = any(ctx.needs_input_grad)
        ldT = (n + 3) // 4 * 4
        packs = (_pack(W1, N1, H + S, False, False), _pack(W2, N2, N1, False, True), _pack(W3, N3, N2, False, True))
        biases = (_pad32(b1), _pad32(b2), _pad32(b3))
        out = torch.empty(n, N3, dtype=torch.float32, device=dev)
        xT = h1T = h2T = None
        if need:
            xT = torch.empty(H + S, ldT, dtype=torch.float32, device=dev)
            h1T = torch.empty(N1, ldT, dtype=torch.float32, device=dev)
            h2T = torch.empty(N2, ldT, dtype=torch.float32, device=dev)
        d = L.MlpChainDesc()
        d.n, d.backward = n, 0
        d.src1, d.ld1, d.gather, d.c1 = ctx2d.data_ptr(), ctx2d.stride(0), idx.data_ptr(), H
        d.src2, d.ld2, d.c2 = x.data_ptr(), x.stride(0), S
        for i in range(3):
            d.packed[i], d.bias[i] = packs[i].data_ptr(), biases[i].data_ptr()
        d.dims[0], d.dims[1], d.dims[2], d.dims[3] = H + S, N1, N2, N3
        d.inT, d.o1T, d.o2T, d.ldT = _p(xT), _p(h1T), _p(h2T), ldT
        d.out, d.ld_out = out.data_ptr(), out.stride(0)
        L.check(lib.egp_mlp_chain_f32(C.byref(d), L.current_stream()), "egp_mlp_chain_f32 (forward)")
        if need:
            ctx.save_for_backward(ctx2d, idx, xT, h1T, h2T, W1, W2, W3)
            ctx.dims = (n, H, S, N1, N2, N3, ldT)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        ctx2d, idx, xT, h1T, h2T, W1, W2, W3 = ctx.saved_tensors
        n, H, S, N1, N2, N3, ldT = ctx.dims
        dev = dout.device
        dout = dout if (dout.stride(1) == 1 or N3 == 1) and dout.dtype == torch.float32 else dout.contiguous().float()
        if N3 == 1 and dout.stride(0) != 1:
            dout = dout.contiguous()
        packs = (_pack(W3, N2, N3, True, False), _pack(W2, N1, N2, True, True), _pack(W1[:, :H], H, N1, True, True))
        dz3T = torch.empty(N3, ldT, dtype=torch.float32, device=dev)
        dz2T = torch.empty(N2, ldT, dtype=torch.float32, device=dev)
        dz1T = torch.empty(N1, ldT, dtype=torch.float32, device=dev)
        dctx = torch.zeros_like(ctx2d)
        d = L.MlpChainDesc()
        d.n, d.backward = n, 1
        d.src1, d.ld1, d.gather, d.c1 = dout.data_ptr(), dout.stride(0), None, N3
        d.src2, d.ld2, d.c2 = None, 0, 0
        for i in range(3):
            d.packed[i] = packs[i].data_ptr()
        d.dims[0], d.dims[1], d.dims[2], d.dims[3] = N3, N2, N1, H
        d.mask1, d.mask2 = h2T.data_ptr(), h1T.data_ptr()
        d.inT, d.o1T, d.o2T, d.ldT = dz3T.data_ptr(), dz2T.data_ptr(), dz1T.data_ptr(), ldT
        d.out, d.ld_out, d.scatter = dctx.data_ptr(), dctx.stride(0), idx.data_ptr()
        L.check(lib.egp_mlp_chain_f32(C.byref(d), L.current_stream()), "egp_mlp_chain_f32 (backward)")
        # weight gradients: dW[out][in] = dz^T[out][rows] . h^T[in][rows] -- both operands k-contiguous (k = batch row)
        def wgrad(dzT, hT):
            A, B = dzT[:, :n], hT[:, :n]
            if A.shape[0] == 1:
                return torch.mv(B, A[0]).unsqueeze(0)
            return G.gemm(A, B, True, True, splits=G.pick_splits(A.shape[0], B.shape[0], n))
        need = ctx.needs_input_grad
        dW1 = wgrad(dz1T, xT) if need[3] else None
        dW2 = wgrad(dz2T, h1T) if need[5] else None
        dW3 = wgrad(dz3T, h2T) if need[7] else None
        db1 = dz1T[:, :n].sum(1) if need[4] else None
        db2 = dz2T[:, :n].sum(1) if need[6] else None
        db3 = dz3T[:, :n].sum(1) if need[8] else None
        return (dctx if need[0] else None), None, None, dW1, db1, dW2, db2, dW3, db3


def chain_mlp_head(gi, layers, head):
    l1, l2 = list(layers)
    return ChainMlpHead.apply(gi.ctx2d, gi.idx, gi.x, l1.weight, l1.bias, l2.weight, l2.bias, head.weight, head.bias)
