"""One LSTM direction with the persistent HIP recurrence (csrc/egp_lstm.hip) behind torch autograd.

Split of the work:
  rocBLAS (MFMA):  the input projection  X W_ih^T + b  for all T*B rows at once, and in backward the three
                   weight-gradient reductions  dW_ih = dPre^T X,  dW_hh = dPre^T H_prev,  db = sum dPre;
  HIP kernels:     the sequential part -- T steps of h W_hh^T + gate non-linearities (forward) and the
                   backward-through-time recurrence producing dPre.
Parameters are those of ``nn.LSTMCell`` (weight_ih [4H,D], weight_hh [4H,H], bias_ih, bias_hh), hidden size 64,
float32, zero initial state: exactly what ``RNN.batch_forward`` of the reference evaluates step by step.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L
from . import gemm as G

HIDDEN_SIZES = (64, 128)      # hidden sizes the HIP kernels are built for


def available(x, cell):
    return (x.is_cuda and x.dtype == torch.float32 and cell.hidden_size in HIDDEN_SIZES
            and cell.weight_hh.dtype == torch.float32 and cell.bias_ih is not None)


def _s():
    return L.current_stream()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_PERM = {}


def _gate_perm(hidden, device):
    """Row permutation W_ih -> the kernels' gate layout (include/egopose_hip.h): unit-major column n = 4*u + g
    takes torch row g*H + u; None when the build uses torch's own order."""
    if L.load().egp_lstm_gate_layout() == 0:
        return None
    key = (hidden, str(device))
    if key not in _PERM:
        n = torch.arange(4 * hidden, device=device)
        perm = (n % 4) * hidden + n // 4
        _PERM[key] = (perm, torch.argsort(perm))
    return _PERM[key]


def _stacked_perm(hidden, P, device):
    """The gate permutation applied to P stacked (4H)-row blocks at once, and its inverse (cached)."""
    key = (hidden, P, str(device))
    if key not in _PERM:
        perm, inv = _gate_perm(hidden, device)
        off = (torch.arange(P, device=device) * 4 * hidden).unsqueeze(1)
        _PERM[key] = ((perm.unsqueeze(0) + off).reshape(-1), (inv.unsqueeze(0) + off).reshape(-1))
    return _PERM[key]


class LstmDirection(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, reverse, train):
        lib = L.load()
        T, B, D = x.shape
        HIDDEN = w_hh.shape[1]
        x2 = x.reshape(T * B, D)
        pp = _gate_perm(HIDDEN, x.device)
        bias = b_ih + b_hh
        w_in = w_ih
        if pp is not None:
            w_in, bias = w_ih.index_select(0, pp[0]), bias.index_select(0, pp[0])
        if G.enabled():                     # bf16 matrix cores, split operands (csrc/egp_gemm.hip)
            gx = G.linear_fwd(x2, w_in.contiguous(), bias).view(T, B, 4 * HIDDEN)
        else:
            gx = torch.addmm(bias, x2, w_in.t()).view(T, B, 4 * HIDDEN)
        # T + 1 time slots with a zero one in front (behind, for the reverse direction): h and the h_prev the weight
        # gradient needs are two views of the same buffer
        h_buf = torch.empty(T + 1, B, HIDDEN, dtype=x.dtype, device=x.device)
        h_buf[T if reverse else 0].zero_()
        h = h_buf[:T] if reverse else h_buf[1:]
        # `train` comes from the caller: ctx.needs_input_grad mirrors requires_grad of the inputs and is also set under
        # torch.no_grad(), where nothing will ever call backward (no cell / gate saves: the inference kernel)
        cells = torch.empty(T, B, HIDDEN, dtype=x.dtype, device=x.device) if train else None
        w_hh_c = w_hh.contiguous()
        # the activated gates overwrite the pre-activations in place (a thread reads its part of gx[t] before it
        # writes the same addresses)
        L.check(lib.egp_lstm_fwd_f32(_p(gx), _p(w_hh_c), T, B, HIDDEN, 1 if reverse else 0, _p(h),
                                     _p(gx if train else None), _p(cells), _s()), "egp_lstm_fwd_f32")
        if train:
            ctx.save_for_backward(x2, w_in, w_hh_c, h_buf, gx, cells)
            ctx.reverse = bool(reverse)
            ctx.shape = (T, B, D)
        return h

    @staticmethod
    def backward(ctx, dh):
        lib = L.load()
        x2, w_in, w_hh, h_buf, gates, cells = ctx.saved_tensors
        T, B, D = ctx.shape
        HIDDEN = w_hh.shape[1]
        h = h_buf
        dpre = torch.empty(T, B, 4 * HIDDEN, dtype=h.dtype, device=h.device)
        L.check(lib.egp_lstm_bwd_f32(_p(dh.contiguous()), _p(gates), _p(cells), _p(w_hh), T, B, HIDDEN,
                                     1 if ctx.reverse else 0, _p(dpre), _s()), "egp_lstm_bwd_f32")
        h_prev = h_buf[1:] if ctx.reverse else h_buf[:T]
        # dW_ih | dW_hh | db in ONE batched GEMM over the time axis + a reduction: [dPre_t^T (x_t | h_prev_t | 1)]
        # (a single (4H x T*B) @ (T*B x D) product runs 3x slower in rocBLAS than T independent ones; separate
        # products per operand would read dPre three times)
        pp = _gate_perm(HIDDEN, h.device)
        if G.enabled():
            # two split-K products over the (time, sequence) rows; the first also returns the bias gradient
            d2 = dpre.view(T * B, 4 * HIDDEN)
            dw_ih, dbias = G.linear_wgrad(d2, x2, want_bias=True)
            dw_hh = G.linear_wgrad(d2, h_prev.reshape(T * B, HIDDEN), want_bias=False)
            if pp is not None:
                dw_ih, dw_hh, dbias = dw_ih.index_select(0, pp[1]), dw_hh.index_select(0, pp[1]), dbias.index_select(0, pp[1])
            d_w_ih = dw_ih if ctx.needs_input_grad[1] else None
            d_w_hh = dw_hh if ctx.needs_input_grad[2] else None
            d_b = dbias if (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]) else None
        else:
            xh1 = torch.cat((x2.view(T, B, D), h_prev, h.new_ones(T, B, 1)), 2)
            dw = torch.bmm(dpre.transpose(1, 2), xh1).sum(0)                   # (4H, D + H + 1), rows in the kernels' gate layout
            if pp is not None:
                dw = dw.index_select(0, pp[1])
            d_w_ih = dw[:, :D].contiguous() if ctx.needs_input_grad[1] else None
            d_w_hh = dw[:, D:D + HIDDEN].contiguous() if ctx.needs_input_grad[2] else None
            d_b = dw[:, D + HIDDEN].contiguous() if (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]) else None
        d_x = dpre.view(T * B, 4 * HIDDEN).mm(w_in).view(T, B, D) if ctx.needs_input_grad[0] else None
        return d_x, d_w_ih, d_w_hh, d_b, d_b, None, None


class LstmGroup(torch.autograd.Function):
    """P sweeps (cells, directions) over the same input x in one grouped launch each way:
    apply(x, reverse_mask, P, width, w_ih_0, w_hh_0, b_ih_0, b_hh_0, w_ih_1, ...).
    width = 1 -> P outputs (T,B,H);  width = 2 -> P/2 outputs (T,B,2H): problems 2i and 2i+1 (the two directions of a
    bi-LSTM) write the halves of one buffer, so no concatenation afterwards and no split of its gradient.
    One projection GEMM against the stacked W_ih; in backward the bias gradient comes out of the kernel, dW_ih of all
    problems is one batched GEMM against x, dW_hh one per problem against its own h_prev (no concatenated operands)."""

    @staticmethod
    def forward(ctx, x, reverse_mask, P, width, train, ragged, frames, *params):
        """`ragged` = None or (order, steps): int32 device tensors of B entries (include/egopose_hip.h,
        egp_lstm_group_fwd_len_f32) -- forward-running problems stop at the longest sequence of their workgroup.
        `frames` = None or (table (F, D), base int32 (B,)) with x[t, b] == table[base[b] + t]: the input projection then runs over
        the table's F rows once and the sweeps read row base[b] + t of it (the reference evaluates W_ih x_t per window row,
        models/rnn.py:45-61; windows of one take overlap, so most of those rows repeat)."""
        lib = L.load()
        T, B, D = x.shape
        w_ih, w_hh, b_ih, b_hh = params[0::4], params[1::4], params[2::4], params[3::4]
        H = w_hh[0].shape[1]
        perm, inv = _gate_perm(H, x.device)
        x2 = x.reshape(T * B, D)
        # problems that run forward in time first: their gate columns are one block (a ragged batch visits fewer rows for them)
        pord = [p for p in range(P) if not (reverse_mask >> p) & 1] + [p for p in range(P) if (reverse_mask >> p) & 1]
        nf = sum(1 for p in range(P) if not (reverse_mask >> p) & 1)
        kmask = sum(1 << q for q in range(nf, P))                                           # reverse bits in kernel order
        # one gather for the P stacked blocks instead of one per problem (the update runs this every epoch)
        bperm, _ = _stacked_perm(H, P, x.device)
        w_in = torch.cat([w_ih[p] for p in pord], 0).index_select(0, bperm)                 # (P*4H, D)
        bias = (torch.cat([b_ih[p] for p in pord]) + torch.cat([b_hh[p] for p in pord])).index_select(0, bperm)
        rows = None
        if ragged is not None and ragged.rows is not None and 0 < nf and G.enabled() and G.fused_rows_available() \
                and ragged.T == T and ragged.rows.shape[0] >= 4096:
            rows = ragged.rows
        seq_base, gates_buf = None, None
        if frames is not None and G.enabled():
            table, seq_base = frames
            gx = G.linear_fwd(table, w_in, bias)                                            # (F, P*4H): one row per frame
            gates_buf = torch.empty(T * B, P * 4 * H, dtype=x.dtype, device=x.device) if train else None
        elif rows is not None:
            gx = torch.empty(T * B, P * 4 * H, dtype=x.dtype, device=x.device)
            nfc = nf * 4 * H
            G.gemm(x2, w_in[:nfc], True, True, bias=bias[:nfc], a_rows=rows, c_rows=rows, out=gx[:, :nfc])
            if nf < P:
                G.gemm(x2, w_in[nfc:], True, True, bias=bias[nfc:], out=gx[:, nfc:])
        elif G.enabled():
            gx = G.linear_fwd(x2, w_in, bias)                                               # (T*B, P*4H)
        else:
            gx = torch.addmm(bias, x2, w_in.t())
        w_hh_all = torch.stack([w_hh[p].contiguous() for p in pord], 0)
        # per output T + 2 time slots, zero | h_0 .. h_{T-1} | zero: h_prev is the same buffer shifted by one slot
        # (down for a forward sweep, up for a reversed one)
        n_out, W = P // width, width * H
        h_buf = torch.empty(n_out, T + 2, B, W, dtype=x.dtype, device=x.device)
        h_buf[:, 0].zero_()
        h_buf[:, T + 1].zero_()
        cells = torch.empty(P, T, B, H, dtype=x.dtype, device=x.device) if train else None
        base, esz = h_buf.data_ptr(), h_buf.element_size()
        ptrs = (C.c_void_p * P)(*[base + esz * (((p // width) * (T + 2) + 1) * B * W + (p % width) * H) for p in pord])
        order, steps = (ragged.order, ragged.steps) if ragged is not None else (None, None)
        # with row lists every later product visits only the rows the workgroups stepped through: the skipped steps of the
        # sweeps (3.6-3.9 TB/s of HBM traffic) need not be filled with zeros
        leave = int(rows is not None and train)
        gates = gates_buf if seq_base is not None else (gx if train else None)             # (a dense projection doubles as the saved gates)
        L.check(lib.egp_lstm_group_fwd_len_f32(_p(gx), _p(w_hh_all), T, B, H, P, kmask, ptrs, W,
                                               _p(gates), _p(cells), _p(order), _p(steps), leave, _p(seq_base), _s()), "egp_lstm_group_fwd_len_f32")
        outs = tuple(h_buf[i, 1:T + 1] for i in range(n_out))
        if train:
            ctx.save_for_backward(x2, w_in, w_hh_all, h_buf, gates, cells)
            ctx.meta = (T, B, D, H, P, width, reverse_mask)
            ctx.ragged, ctx.rows, ctx.pord, ctx.nf, ctx.kmask, ctx.leave = ragged, rows, pord, nf, kmask, leave
        return outs

    @staticmethod
    def backward(ctx, *douts):
        lib = L.load()
        x2, w_in, w_hh_all, h_buf, gates, cells = ctx.saved_tensors
        T, B, D, H, P, width, reverse_mask = ctx.meta
        W = width * H
        perm, inv = _gate_perm(H, x2.device)
        douts = [d.contiguous() if d is not None else h_buf.new_zeros(T, B, W) for d in douts]
        ragged, rows, pord, nf, kmask = ctx.ragged, ctx.rows, ctx.pord, ctx.nf, ctx.kmask
        esz = h_buf.element_size()
        ptrs = (C.c_void_p * P)(*[douts[p // width].data_ptr() + esz * (p % width) * H for p in pord])
        dpre = torch.empty(T * B, P * 4 * H, dtype=x2.dtype, device=x2.device)
        db = torch.zeros(P, 4 * H, dtype=x2.dtype, device=x2.device)
        order, steps = (ragged.order, ragged.steps) if ragged is not None else (None, None)
        leave = int(bool(ctx.leave) and not ctx.needs_input_grad[0])     # (d_x = d_pre W_in reads every row)
        L.check(lib.egp_lstm_group_bwd_len_f32(ptrs, W, _p(gates), _p(cells), _p(w_hh_all), T, B, H, P, kmask, _p(dpre), _p(db),
                                               _p(order), _p(steps), leave, _s()), "egp_lstm_group_bwd_len_f32")
        d3 = dpre.view(T, B, P * 4 * H)
        use_g = G.enabled()
        nfc = nf * 4 * H
        if rows is not None:      # forward-running problems: the rows their workgroups stepped through; the others: every row
            n_rows = rows.shape[0]
            parts = [G.gemm(dpre[:, :nfc], x2, False, False, splits=G.pick_splits(nfc, D, n_rows), a_krows=rows, b_krows=rows)]
            if nf < P:
                parts.append(G.linear_wgrad(dpre[:, nfc:], x2, want_bias=False))
            dw_ih_all = torch.cat(parts, 0)
        elif use_g:      # one split-K product for the P stacked W_ih gradients
            dw_ih_all = G.linear_wgrad(dpre, x2, want_bias=False)                          # (P*4H, D), kernel gate order
        else:
            dw_ih_all = torch.bmm(d3.transpose(1, 2), x2.view(T, B, D)).sum(0)
        grads = [None] * (4 * P)
        _, binv = _stacked_perm(H, P, x2.device)
        dw_ih_t = dw_ih_all.index_select(0, binv)                                           # torch row order, P blocks at once
        db_t = db.reshape(-1).index_select(0, binv).view(P, 4 * H)
        for q, p in enumerate(pord):          # q: the problem's place in the kernels' order, p: in the caller's
            rev = (kmask >> q) & 1
            slab = h_buf[p // width, 2:] if rev else h_buf[p // width, :T]
            h_prev = slab[:, :, (p % width) * H:(p % width + 1) * H]
            if use_g:  # strided views: problem q's columns of d_pre against its half of the shifted hidden buffer
                dq, hq = dpre[:, q * 4 * H:(q + 1) * 4 * H], slab.reshape(T * B, W)[:, (p % width) * H:(p % width + 1) * H]
                if rows is not None and not rev:
                    dw_hh = G.gemm(dq, hq, False, False, splits=G.pick_splits(4 * H, H, rows.shape[0]), a_krows=rows, b_krows=rows).index_select(0, inv)
                else:
                    dw_hh = G.linear_wgrad(dq, hq, want_bias=False).index_select(0, inv)
            else:
                dw_hh = torch.bmm(d3[:, :, q * 4 * H:(q + 1) * 4 * H].transpose(1, 2), h_prev).sum(0).index_select(0, inv)
            dw_ih = dw_ih_t[q * 4 * H:(q + 1) * 4 * H]
            d_b = db_t[q]
            grads[4 * p:4 * p + 4] = [dw_ih, dw_hh, d_b, d_b]
        d_x = dpre.mm(w_in).view(T, B, D) if ctx.needs_input_grad[0] else None
        return (d_x, None, None, None, None, None, None, *grads)


def _wants_grad(x, params):
    """True when a backward pass can follow: grad mode on and something to differentiate. Decided outside the autograd
    Function (inside its forward grad mode is always off, and needs_input_grad ignores torch.no_grad())."""
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))


def group_available(x, cells):
    """Grouped sweeps need the matrix-core kernels, at most four problems and identical cell shapes."""
    if not cells or len(cells) > 4 or L.load().egp_lstm_gate_layout() == 0:
        return False
    c0 = cells[0]
    return all(available(x, c) and c.hidden_size == c0.hidden_size and c.input_size == c0.input_size for c in cells)


class Ragged:
    """What the grouped sweeps need to know about a ragged batch (ragged_order)."""

    def __init__(self, order, steps, rows, n_seq, T):
        self.order, self.steps, self.rows, self.n_seq, self.T = order, steps, rows, n_seq, T


def ragged_order(seq_steps, device, T=None):
    """Ragged (order, steps[, rows]) for LstmGroup from the time steps each sequence needs (B entries, in layout order):
    positions sorted by decreasing length, so that the sequences of a workgroup are alike. With the window length `T`
    also `rows`: the (t, b) rows -- flattened t * B + b, ascending -- that forward-running workgroups step through
    (a workgroup of up to 8 positions runs to its longest sequence); the input projection and the weight gradients of the
    forward-running problems visit only these."""
    st = np.asarray(torch.as_tensor(seq_steps).cpu().numpy(), dtype=np.int64).reshape(-1)
    B = st.shape[0]
    order = np.argsort(-st, kind="stable")
    st_sorted = st[order]
    rows = None
    if T is not None and B > 0:
        wg = np.maximum.reduceat(st_sorted, np.arange(0, B, 8))              # positions are sorted: the first of each group of 8
        t_wg = np.minimum(np.repeat(wg, 8)[:B], int(T))                      # steps the workgroup of position p runs
        n = int(t_wg.sum())
        if n < 0.9 * T * B:                                                  # otherwise not worth the indirection
            t_seq = np.empty(B, np.int64)
            t_seq[order] = t_wg                                              # steps of the workgroup that sequence b sits in
            stepped = np.arange(int(T), dtype=np.int64)[:, None] < t_seq[None, :]      # (T, B): row t * B + b is stepped through
            rows = torch.as_tensor(np.flatnonzero(stepped), dtype=torch.int64, device=device)     # ascending, no sort
    return Ragged(torch.as_tensor(order, dtype=torch.int32, device=device), torch.as_tensor(st_sorted, dtype=torch.int32, device=device),
                  rows, B, T)


def lstm_group(x, cells, reverses, pairs=False, ragged=None, frames=None):
    """P (cell, reverse) pairs over the same x (T,B,D), one grouped launch each way. Returns [(T,B,H)] * P, or with
    pairs=True [(T,B,2H)] * P/2 where problems 2i and 2i+1 fill the two halves of output i. `ragged` (ragged_order(...)):
    outputs of a forward-running problem beyond a sequence's own steps are not computed (zeros) and carry no gradient.
    `frames` = (table (F, D), base int32 (B,)) when x[t, b] == table[base[b] + t] (windows of consecutive frames): the input
    projection is then computed once per frame of the table (LstmGroup.forward); x itself still serves the weight gradient."""
    mask = sum(1 << i for i, r in enumerate(reverses) if r)
    params = [t for c in cells for t in (c.weight_ih, c.weight_hh, c.bias_ih, c.bias_hh)]
    if ragged is not None and ragged.n_seq != x.shape[1]:
        ragged = None
    if frames is not None:
        table, base = frames
        ok = (table.is_cuda and table.dtype == x.dtype and table.dim() == 2 and table.shape[1] == x.shape[2] and table.is_contiguous()
              and base.dtype == torch.int32 and base.is_cuda and base.is_contiguous() and base.shape[0] == x.shape[1] and not x.requires_grad)
        if not ok:
            frames = None
    return list(LstmGroup.apply(x.contiguous(), mask, len(cells), 2 if pairs else 1, _wants_grad(x, params), ragged, frames, *params))


def lstm_direction(cell, x, reverse):
    params = (cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
    return LstmDirection.apply(x, *params, bool(reverse), _wants_grad(x, params))
