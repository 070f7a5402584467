"""UMPR on MI355X: the reference's ``src/model.py`` module tree (same constructor, same ``forward`` signature,
same state_dict keys) whose arithmetic runs in hand-written gfx950 kernels behind the C ABI of libumpr_hip.so.

The torch ``nn`` modules below (nn.GRU, nn.Conv1d, nn.Linear, ...) are parameter holders only: they give the
reference's parameter names, shapes and default initialisation; their ``forward`` is never called.  Each
``torch.autograd.Function`` is one fused region (K-numbers: SURVEY.md section 2a) and calls the C ABI for forward
and backward.  There is no torch/CPU fallback - without the library or a GPU the model raises.

Reference lines: UMPR.__init__ src/model.py:233-255, UMPR.forward src/model.py:257-278.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading

import torch
from torch import nn

from . import _lib as _lib_mod
from ._lib import Workspace, lib, stream_ptr
from .streams import note_gradients_written

TEXT_STREAM = os.environ.get("UMPR_TEXT_STREAM", "1") != "0"   # text path on side streams beside the VGG stack
# (UMPR_TEXT_STREAMS=2 - ReviewNet and ControlNet on streams of their own - measured slower in round 2 and went away with the fused
# text path of round 3: both are issued back to back on the one side stream)
_SIDE_STREAMS = {}
_TEXT_TLS = threading.local()     # .on_side: the text Functions below are being issued on the side stream (UMPR._forward)
H = 64          # config.gru_size the kernels are built for
D = 2 * H
AT = 64         # config.self_atte_size


def _ws(nbytes, device):
    buf = Workspace.get(max(int(nbytes), 256), device)
    return buf, buf.numel() * 4


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _grad_targets(params, can_accumulate=False):
    """Where a backward node writes each parameter gradient: straight into ``p.grad`` (a view of FusedAdam's flat
    arena) when the optimiser marked it fresh in ``zero_grad()`` - no zero fill, no temporary, no accumulate kernel -
    else into a new tensor that autograd accumulates as usual (second backward without zero_grad, plain optimisers).
    ``can_accumulate``: the kernel can ADD onto existing gradients, so a parameter another node already wrote in this
    step (shared weights: one GRU serves several review tensors) is also updated in place.  Returns (targets, direct,
    accumulate) with ``accumulate`` true when the kernel must add; all parameters of a call share one mode."""
    def usable(p):
        if not p.is_leaf:        # a zero-padded copy of a parameter (_ZeroPad): autograd carries its gradient back
            return False
        g = p.grad
        return (g is not None and getattr(p, "_umpr_direct", False) and g.is_contiguous() and g.shape == p.shape
                and g.device == p.device and g.dtype == torch.float32)
    fresh = [usable(p) and getattr(p, "_umpr_fresh", False) for p in params]
    written = [usable(p) and getattr(p, "_umpr_written", False) and not getattr(p, "_umpr_fresh", False) for p in params]
    if can_accumulate and params and all(written):
        return [p.grad for p in params], [True] * len(params), True
    if can_accumulate and not all(fresh):
        fresh = [False] * len(params)          # mixed state: let autograd accumulate everything
    dst = [p.grad if ok else torch.empty_like(p) for p, ok in zip(params, fresh)]
    return (dst, fresh, False) if can_accumulate else (dst, fresh)


def _grad_returns(params, dst, direct):
    for p, d in zip(params, direct):
        if d:
            p._umpr_fresh = False
            p._umpr_written = True    # holds this step's gradient: a later node of the same step may add in place
    return [None if d else g for g, d in zip(dst, direct)]


# --------------------------------------------------------------------------------------------- K1-K3
# Mixed precision of the text path's GEMM-shaped products (bf16 mode only; UMPR_TEXT_BF16=0 keeps them on the fp32 MFMA
# kernels): UMPR.forward publishes the mode, every text Function records it at forward time and brackets its library calls
# with umpr_set_gemm_bf16 (through _lib.TLS.gemm_b16) in forward AND in its backward, whichever thread autograd runs that on.
_TEXT_BF16 = os.environ.get("UMPR_TEXT_BF16", "1") != "0"


class _Mode(threading.local):
    """Per-thread forward state (with the reference's thread-per-replica caller, forward must be re-entrant -
    no process-global mutable state).  b16: the text Functions of the forward in progress run their products on the bf16
    pipe; infer: the VGG forward in progress runs under torch.no_grad() (VGG16.forward)."""
    b16 = False
    infer = False


_MODE = _Mode()


class _b16_products:
    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _lib_mod.TLS.gemm_b16
        _lib_mod.TLS.gemm_b16 = self.on

    def __exit__(self, *exc):
        _lib_mod.TLS.gemm_b16 = self.prev


def _b16_forward(f):
    def w(ctx, *a):
        ctx.b16 = _MODE.b16
        with _b16_products(ctx.b16):
            return f(ctx, *a)
    return staticmethod(w)


def _b16_backward(f):
    def w(ctx, *a):
        with _b16_products(ctx.b16):
            return f(ctx, *a)
    return staticmethod(w)


class _EmbedGru(torch.autograd.Function):
    """nn.Embedding + ImprovedRnn(nn.GRU bidirectional) (src/model.py:262-264, 12-21).  ``split`` > 0: the batch is two
    review tensors of ``split`` sequences each run in one launch (UMPR._pair); their outputs come back as two tensors
    (views of one buffer) and their gradients are joined by one concatenation instead of autograd's slice bookkeeping."""

    @_b16_forward
    def forward(ctx, ids, lengths, order, emb, split, *w):
        N, L = ids.shape
        E = emb.shape[1]
        need = any(ctx.needs_input_grad)
        out = torch.empty(N, L, D, device=ids.device, dtype=torch.float32)
        saved = torch.empty(2, N, L, 4, H, device=ids.device, dtype=torch.float32) if need else None
        ws, wsb = _ws(lib().size("umpr_embed_gru_bidir_ws_bytes", N, L, E), ids.device)
        ctx.param_objs = w
        ctx.split = int(split)
        w = [_c(x) for x in w]
        lib().call("umpr_embed_gru_bidir_fwd", ids, emb, E, *w, lengths, order, order, N, L, out, saved, ws, wsb,
                   stream_ptr())
        if need:
            ctx.save_for_backward(ids, lengths, order, emb, w[1], w[5], saved)
            ctx.out = out        # the kernel's own output buffer (with split, the returned tensors are views of it)
        if split:
            return out[:split], out[split:]
        return out

    @_b16_backward
    def backward(ctx, *douts):
        ids, lengths, order, emb, whh_f, whh_r, saved = ctx.saved_tensors
        out = ctx.out
        if out is None:   # ctx.out is dropped after the first backward (it would otherwise pin the output buffer)
            raise RuntimeError("_EmbedGru: backward was already run for this forward (the GRU output buffer is released "
                               "after the first backward; run forward again instead of retain_graph=True)")
        N, L = ids.shape
        E = emb.shape[1]
        dev = ids.device
        if ctx.split:
            parts = [d if d is not None else torch.zeros(n, L, D, device=dev)
                     for d, n in zip(douts, (ctx.split, N - ctx.split))]
            dout = torch.cat(parts)
        else:
            dout = _c(douts[0])
        # the reference shares one GRU between the user and item reviews (and one between the three C-Net calls): the first
        # backward of a step overwrites the gradient slices of the optimiser's arena, the later ones add in place
        g, direct, acc = _grad_targets(ctx.param_objs, can_accumulate=True)
        ws, wsb = _ws(lib().size("umpr_embed_gru_bidir_ws_bytes", N, L, E), dev)
        lib().call("umpr_embed_gru_bidir_bwd_acc", ids, emb, E, whh_f, whh_r, lengths, order, order, N, L, dout, out,
                   saved, *g, int(acc), ws, wsb, stream_ptr())
        ctx.out = None
        return (None, None, None, None, None, *_grad_returns(ctx.param_objs, g, direct))


# --------------------------------------------------------------------------------------------- K4-K7
class _ReviewHead(torch.autograd.Function):
    """R-Net co-attention + S-Net(u) + S-Net(i) + textual matching (src/model.py:50-55, 71-81, 162-168)."""

    @_b16_forward
    def forward(ctx, gru_u, gru_i, S, L, M, Ms_u, Ws_u, Ms_i, Ws_i, W_u, W_i, bf16=False):
        B, SL, _ = gru_u.shape
        dev = gru_u.device
        f = dict(device=dev, dtype=torch.float32)
        ctx.param_objs = (M, Ms_u, Ws_u, Ms_i, Ws_i, W_u, W_i)
        gru_u, gru_i = _c(gru_u), _c(gru_i)
        T = torch.empty(B, SL, D, **f)
        soft_u, soft_i = torch.empty(B, SL, **f), torch.empty(B, SL, **f)
        colmax, rowmax = torch.empty(B, SL, **f), torch.empty(B, SL, **f)
        argcol = torch.empty(B, SL, device=dev, dtype=torch.int32)
        argrow = torch.empty(B, SL, device=dev, dtype=torch.int32)
        repr_u, repr_i = torch.empty(B, 2 * D, **f), torch.empty(B, 2 * D, **f)
        ws, wsb = _ws(lib().size("umpr_coattention_fwd_ws_bytes", B, SL), dev)
        st = stream_ptr()
        lib().call("umpr_coattention_fwd_bf16" if bf16 else "umpr_coattention_fwd", gru_u, gru_i, M, B, SL, T, soft_u,
                   soft_i, repr_u, 2 * D, repr_i, 2 * D, colmax, argcol, rowmax, argrow, ws, wsb, st)
        sn = []
        for X, Ms, Ws, soft, rep in ((gru_u, Ms_u, Ws_u, soft_u, repr_u), (gru_i, Ms_i, Ws_i, soft_i, repr_i)):
            U = torch.empty(B, S, L, AT, **f)
            P = torch.empty(B, S, L, **f)
            wsum = torch.empty(B, S, **f)
            sa = torch.empty(B, S, D, **f)
            lib().call("umpr_snet_fwd", X, Ms, Ws, soft, L, B, S, L, U, P, wsum, sa, rep.data_ptr() + D * 4, 2 * D, st)
            sn += [U, P, wsum, sa]
        out = torch.empty(B, D, **f)
        lib().call("umpr_review_merge_fwd", repr_u, repr_i, W_u, W_i, B, out, st)
        ctx.dims = (B, S, L)
        ctx.save_for_backward(gru_u, gru_i, M, Ms_u, Ws_u, Ms_i, Ws_i, W_u, W_i, T, soft_u, soft_i, colmax, argcol,
                              rowmax, argrow, repr_u, repr_i, out, *sn)
        return out

    @_b16_backward
    def backward(ctx, d_out):
        (gru_u, gru_i, M, Ms_u, Ws_u, Ms_i, Ws_i, W_u, W_i, T, soft_u, soft_i, colmax, argcol, rowmax, argrow, repr_u,
         repr_i, out, U_u, P_u, wsum_u, sa_u, U_i, P_i, wsum_i, sa_i) = ctx.saved_tensors
        B, S, L = ctx.dims
        SL = S * L
        dev = gru_u.device
        f = dict(device=dev, dtype=torch.float32)
        st = stream_ptr()
        d_repr_u, d_repr_i = torch.empty(B, 2 * D, **f), torch.empty(B, 2 * D, **f)
        tg, direct = _grad_targets(ctx.param_objs)    # (M, Ms_u, Ws_u, Ms_i, Ws_i, W_u, W_i): each used once per step
        dW_u, dW_i = tg[5], tg[6]
        ws, wsb = _ws(lib().size("umpr_review_merge_bwd_ws_bytes", B), dev)
        lib().call("umpr_review_merge_bwd", repr_u, repr_i, W_u, W_i, out, _c(d_out), B, d_repr_u, d_repr_i, dW_u, dW_i,
                   ws, wsb, st)
        dG, dMs, dWs, dsoft = [], [], [], []
        ws, wsb = _ws(lib().size("umpr_snet_bwd_ws_bytes", B, S, L), dev)
        for X, Ms, Ws, U, P, wsum, sa, drep, gMs, gWs in ((gru_u, Ms_u, Ws_u, U_u, P_u, wsum_u, sa_u, d_repr_u, tg[1], tg[2]),
                                                          (gru_i, Ms_i, Ws_i, U_i, P_i, wsum_i, sa_i, d_repr_i, tg[3], tg[4])):
            dX = torch.empty(B, SL, D, **f)
            ds = torch.empty(B, SL, **f)
            lib().call("umpr_snet_bwd", X, Ms, Ws, U, P, wsum, sa, drep.data_ptr() + D * 4, 2 * D, None, B, S, L, L, dX,
                       gMs, gWs, ds, ws, wsb, st)
            dG.append(dX); dMs.append(gMs); dWs.append(gWs); dsoft.append(ds)
        dM = tg[0]
        ws, wsb = _ws(lib().size("umpr_coattention_bwd_ws_bytes", B, SL), dev)
        lib().call("umpr_coattention_bwd", gru_u, gru_i, M, T, soft_u, soft_i, colmax, argcol, rowmax, argrow,
                   d_repr_u, 2 * D, d_repr_i, 2 * D, dsoft[0], dsoft[1], B, SL, dG[0], dG[1], dM, 1, ws, wsb, st)
        ret = _grad_returns(ctx.param_objs, tg, direct)
        return (dG[0], dG[1], None, None, *ret, None)


# --------------------------------------------------------------------------------------------- K8-K9
class _Control(torch.autograd.Function):
    """C-Net heads on (ui, user, item) + control S-Net + SS-Net gate (src/model.py:118-125, 179-198)."""

    @_b16_forward
    def forward(ctx, g_ui, g_u, g_i, dims, thr, Wc, bc, Wl, bl, Ms, Ws, ssW, ssb):
        B, S_ui, L_ui, S, L = dims
        KC, _, KS = Wc.shape
        V = Wl.shape[0]
        dev = g_ui.device
        f = dict(device=dev, dtype=torch.float32)
        st = stream_ptr()
        ctx.param_objs = (Wc, bc, Wl, bl, Ms, Ws, ssW, ssb)
        g_ui, g_u, g_i = _c(g_ui), _c(g_u), _c(g_i)
        saved = []
        finals = []
        for X, s, l in ((g_ui, S_ui, L_ui), (g_u, S, L), (g_i, S, L)):
            Y = torch.empty(B, s, l, KC, **f)
            cmax = torch.empty(B, s, KC, **f)
            argl = torch.empty(B, s, KC, device=dev, dtype=torch.int32)
            sp, vp = torch.empty(B, s, V, **f), torch.empty(B, s, V, **f)
            fin = torch.empty(B, V, **f)
            ws, wsb = _ws(lib().size("umpr_cnet_head_fwd_ws_bytes", B, s, l, KS), dev)
            lib().call("umpr_cnet_head_fwd", X, Wc, bc, Wl, bl, float(thr), B, s, l, KC, KS, V, Y, cmax, argl, sp, vp,
                       fin, ws, wsb, st)
            saved += [cmax, argl, sp, vp]
            finals.append(fin)
        view_p, c_out = saved[3], finals[0]
        U = torch.empty(B, S_ui, L_ui, AT, **f)
        P = torch.empty(B, S_ui, L_ui, **f)
        wsum = torch.empty(B, S_ui, **f)
        sa = torch.empty(B, S_ui, D, **f)
        senti_unused = torch.empty(B, D, **f)
        lib().call("umpr_snet_fwd", g_ui, Ms, Ws, view_p, V, B, S_ui, L_ui, U, P, wsum, sa, senti_unused, D, st)
        senti, vs = torch.empty(B, S_ui, **f), torch.empty(B, V, **f)
        pp, pn = torch.empty(B, V, **f), torch.empty(B, V, **f)
        lib().call("umpr_control_gate_fwd", sa, ssW, ssb, view_p, c_out, B, S_ui, V, senti, vs, pp, pn, st)
        ctx.dims = dims
        ctx.save_for_backward(g_ui, g_u, g_i, Wc, Wl, Ms, Ws, ssW, c_out, U, P, wsum, sa, senti, vs, *saved)
        return finals[1], finals[2], pp, pn

    @_b16_backward
    def backward(ctx, d_cu, d_ci, d_pp, d_pn):
        (g_ui, g_u, g_i, Wc, Wl, Ms, Ws, ssW, c_out, U, P, wsum, sa, senti, vs, *saved) = ctx.saved_tensors
        B, S_ui, L_ui, S, L = ctx.dims
        KC, _, KS = Wc.shape
        V = Wl.shape[0]
        dev = g_ui.device
        f = dict(device=dev, dtype=torch.float32)
        st = stream_ptr()
        zeros = lambda *s: torch.zeros(*s, **f)
        d_cu = _c(d_cu) if d_cu is not None else zeros(B, V)
        d_ci = _c(d_ci) if d_ci is not None else zeros(B, V)
        d_pp = _c(d_pp) if d_pp is not None else zeros(B, V)
        d_pn = _c(d_pn) if d_pn is not None else zeros(B, V)
        view_p = saved[3]
        d_sa = torch.empty(B, S_ui, D, **f)
        d_vp = torch.empty(B, S_ui, V, **f)
        d_cout = torch.empty(B, V, **f)
        tg, direct = _grad_targets(ctx.param_objs)    # (Wc, bc, Wl, bl, Ms, Ws, ssW, ssb): this node is their only writer
        dssW, dssb = tg[6], tg[7]
        ws, wsb = _ws(lib().size("umpr_control_gate_bwd_ws_bytes", B), dev)
        lib().call("umpr_control_gate_bwd", sa, ssW, view_p, c_out, senti, vs, d_pp, d_pn, B, S_ui, V, d_sa, d_vp,
                   d_cout, dssW, dssb, ws, wsb, st)
        dX_ui = torch.empty(B, S_ui * L_ui, D, **f)
        dMs, dWs = tg[4], tg[5]
        ws, wsb = _ws(lib().size("umpr_snet_bwd_ws_bytes", B, S_ui, L_ui), dev)
        lib().call("umpr_snet_bwd", g_ui, Ms, Ws, U, P, wsum, sa, zeros(B, D), D, d_sa, B, S_ui, L_ui, V, dX_ui, dMs,
                   dWs, None, ws, wsb, st)
        dWc, dbc, dWl, dbl = tg[0], tg[1], tg[2], tg[3]
        dX_u, dX_i = torch.empty(B, S * L, D, **f), torch.empty(B, S * L, D, **f)
        calls = ((g_ui, S_ui, L_ui, saved[0:4], d_cout, d_vp, dX_ui, 1, 0),
                 (g_u, S, L, saved[4:8], d_cu, None, dX_u, 0, 1),
                 (g_i, S, L, saved[8:12], d_ci, None, dX_i, 0, 1))
        for X, s, l, (cmax, argl, sp, vp), dfin, dvp, dX, accx, accw in calls:
            ws, wsb = _ws(lib().size("umpr_cnet_head_bwd_ws_bytes", B, s, l, KC, KS, V), dev)
            lib().call("umpr_cnet_head_bwd", X, Wc, Wl, cmax, argl, sp, vp, dfin, dvp, B, s, l, KC, KS, V, dX, accx,
                       accw, dWc, dbc, dWl, dbl, ws, wsb, st)
        ret = _grad_returns(ctx.param_objs, tg, direct)
        return (dX_ui, dX_u, dX_i, None, None, *ret)


# --------------------------------------------------------------------------------------------- smaller hidden sizes
# config.gru_size / self_atte_size below the kernels' 64 (config.py:34-35 expose both; src/model.py:26-29,61-64,148-155 is generic
# in them): the h-unit model is EXACTLY the 64-unit model whose extra units have all-zero parameters - a GRU unit with zero
# weights and biases has r = z = 1/2, n = tanh(0) = 0 and stays at h' = z h = 0 from h0 = 0, contributes 0 to every product
# downstream, and a sum is not changed by adding zeros.  So the reference-shaped parameters (state_dict compatible) are scattered
# into zero tensors of the kernels' shapes on the way in (_ZeroPad: a fill and a copy, no arithmetic) and their gradients gathered
# back on the way out; predictions, loss and gradients equal the oracle's at the configured sizes
# (tests/test_gpu_parity.py::test_hidden_sizes_below_the_kernel_width_vs_oracle).  Costs the 64-wide work; sizes above 64 are
# refused.
class _ZeroPad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, shape, idx0, idx1):
        out = torch.zeros(shape, device=x.device, dtype=x.dtype)
        if idx1 is None:
            out.index_copy_(0, idx0, x)
        elif idx0 is None:
            out.index_copy_(1, idx1, x)
        else:
            rows = torch.zeros((x.shape[0],) + tuple(shape[1:]), device=x.device, dtype=x.dtype)
            rows.index_copy_(1, idx1, x)
            out.index_copy_(0, idx0, rows)
        ctx.idx = (idx0, idx1)
        return out

    @staticmethod
    def backward(ctx, g):
        idx0, idx1 = ctx.idx
        if idx0 is not None:
            g = g.index_select(0, idx0)
        if idx1 is not None:
            g = g.index_select(1, idx1)
        return g, None, None, None


class _PadMaps:
    """Index maps from the configured sizes (h = gru_size, at = self_atte_size) into the kernels' (64, 64)."""

    def __init__(self, h, at, n_views, device):
        ar = lambda n, off=0: torch.arange(n, device=device) + off
        self.gate = torch.cat([ar(h, q * H) for q in range(3)])            # 3h gate rows -> 192 (gate order r, z, n)
        self.hid = ar(h)                                                    # h -> 64
        self.rep = torch.cat([ar(h), ar(h, H)])                             # 2h ([forward; backward]) -> 128
        self.rep2 = torch.cat([self.rep, self.rep + D])                     # [atte; senti] 4h -> 256
        self.att = ar(at)                                                   # at -> 64
        self.fus = torch.cat([self.rep, ar(2 * n_views, D)])                # linear_fusion input 2h (+ 2V) -> 128 (+ 2V)


# --------------------------------------------------------------------------------------------- fused text path
# UMPR.forward issues the text path through these two Functions: ONE C call per direction each (csrc/text_path.hip), one arena
# tensor saved for backward.  The stage-level Functions above remain the unit the parity tests pin kernel by kernel.
class _ReviewNetF(torch.autograd.Function):
    """The whole ReviewNet (src/model.py:157-169) on the user+item pair: params = the R-Net GRU's eight, M, Ms_u, Ws_u, Ms_i,
    Ws_i, W_u, W_i."""

    @staticmethod
    def forward(ctx, ids_pair, lens, order, emb, dims, b16_gemm, b16_scores, *params):
        B, S, L = dims
        E = emb.shape[1]
        dev = ids_pair.device
        need = any(ctx.needs_input_grad)
        arena = torch.empty(lib().size("umpr_review_net_arena_bytes", B, S, L) // 4, device=dev, dtype=torch.float32)
        out = torch.empty(B, D, device=dev, dtype=torch.float32)
        ws, wsb = _ws(lib().size("umpr_review_net_ws_bytes", B, S, L, E), dev)
        cp = [_c(p) for p in params]
        keep, parr = _ptr_array(cp)
        lib().call("umpr_review_net_fwd", ids_pair, emb, E, parr, lens, order, B, S, L, int(b16_gemm), int(b16_scores), int(need),
                   arena, out, ws, wsb, stream_ptr())
        ctx.param_objs = params
        ctx.meta = (B, S, L, int(b16_gemm))
        ctx.on_side = getattr(_TEXT_TLS, "on_side", False)
        if need:
            ctx.save_for_backward(ids_pair, lens, order, emb, arena, *cp)
        return out

    @staticmethod
    def backward(ctx, d_out):
        ids_pair, lens, order, emb, arena, *cp = ctx.saved_tensors
        B, S, L, b16 = ctx.meta
        E = emb.shape[1]
        tg, direct = _grad_targets(ctx.param_objs)     # every parameter of the ReviewNet is used by this node only
        ws, wsb = _ws(lib().size("umpr_review_net_ws_bytes", B, S, L, E), ids_pair.device)
        keep_p, parr = _ptr_array(cp)
        keep_g, garr = _ptr_array(tg)
        lib().call("umpr_review_net_bwd", ids_pair, emb, E, parr, lens, order, B, S, L, b16, arena, _c(d_out), garr, ws, wsb,
                   stream_ptr())
        if ctx.on_side and any(direct):
            note_gradients_written(ids_pair.device)     # in-place gradients from the side stream: umpr_amd/streams.py
        return (None, None, None, None, None, None, None, *_grad_returns(ctx.param_objs, tg, direct))


class _ControlNetF(torch.autograd.Function):
    """The whole ControlNet (src/model.py:179-198): params = the C-Net GRU's eight, cnn weight / bias, linear weight / bias,
    control S-Net Ms / Ws, SS-Net weight / bias.  Returns (c_u, c_i, prefer_pos, prefer_neg)."""

    @staticmethod
    def forward(ctx, ids_ui, ids_pair, lens_ui, ord_ui, lens, order, emb, dims, thr, b16_gemm, *params):
        B, S_ui, L_ui, S, L = dims
        E = emb.shape[1]
        KC, _, KS = params[8].shape
        V = params[10].shape[0]
        dev = ids_pair.device
        need = any(ctx.needs_input_grad)
        arena = torch.empty(lib().size("umpr_control_net_arena_bytes", B, S_ui, L_ui, S, L, KC, V) // 4, device=dev,
                            dtype=torch.float32)
        outs = torch.empty(4, B, V, device=dev, dtype=torch.float32)
        ws, wsb = _ws(lib().size("umpr_control_net_ws_bytes", B, S_ui, L_ui, S, L, E, KC, KS, V), dev)
        cp = [_c(p) for p in params]
        keep, parr = _ptr_array(cp)
        lib().call("umpr_control_net_fwd", ids_ui, ids_pair, emb, E, parr, lens_ui, ord_ui, lens, order, B, S_ui, L_ui, S, L, KC, KS, V,
                   float(thr), int(b16_gemm), int(need), arena, outs[0], outs[1], outs[2], outs[3], ws, wsb, stream_ptr())
        ctx.param_objs = params
        ctx.meta = (B, S_ui, L_ui, S, L, KC, KS, V, int(b16_gemm))
        ctx.on_side = getattr(_TEXT_TLS, "on_side", False)
        ctx.set_materialize_grads(False)
        if need:
            ctx.save_for_backward(ids_ui, ids_pair, lens_ui, ord_ui, lens, order, emb, arena, *cp)
        return outs[0], outs[1], outs[2], outs[3]

    @staticmethod
    def backward(ctx, d_cu, d_ci, d_pp, d_pn):
        ids_ui, ids_pair, lens_ui, ord_ui, lens, order, emb, arena, *cp = ctx.saved_tensors
        B, S_ui, L_ui, S, L, KC, KS, V, b16 = ctx.meta
        E = emb.shape[1]
        dev = ids_pair.device
        d = [(_c(t) if t is not None else torch.zeros(B, V, device=dev)) for t in (d_cu, d_ci, d_pp, d_pn)]
        tg, direct = _grad_targets(ctx.param_objs)
        ws, wsb = _ws(lib().size("umpr_control_net_ws_bytes", B, S_ui, L_ui, S, L, E, KC, KS, V), dev)
        keep_p, parr = _ptr_array(cp)
        keep_g, garr = _ptr_array(tg)
        lib().call("umpr_control_net_bwd", ids_ui, ids_pair, emb, E, parr, lens_ui, ord_ui, lens, order, B, S_ui, L_ui, S, L, KC, KS,
                   V, b16, arena, d[0], d[1], d[2], d[3], garr, ws, wsb, stream_ptr())
        if ctx.on_side and any(direct):
            note_gradients_written(dev)
        return (None,) * 10 + tuple(_grad_returns(ctx.param_objs, tg, direct))


# --------------------------------------------------------------------------------------------- K10
# Data parallel: callables(block) run on the host each time the feature backward has enqueued one VGG block (4 .. 0) - the
# gradient reducer starts that block's all-reduce from there (parallel.GradReducer).  Only when the gradients are written
# in place into the optimiser's arena, i.e. when the slice the hook exchanges really holds this step's values.
# The hook is the attribute `_umpr_block_hook` (a weakref.WeakMethod) of the first conv weight Parameter of the VGG16 being
# differentiated: it lives and dies with that parameter / with the reducer that set it (parallel.GradReducer).
_BLOCK_CB_TYPE = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_void_p)


def _features_bwd_call(name, direct, params, *args):
    ref = getattr(params[0], "_umpr_block_hook", None) if params else None
    hook = ref() if ref is not None else None
    if hook is not None and all(direct):
        failed = []

        def _cb(block, _user):
            # ctypes prints and then DROPS an exception raised inside a C callback: the backward would carry on and this rank
            # would skip a collective its peers issue.  Keep the first one and re-raise it once the library call returns.
            if failed:
                return
            try:
                hook(int(block))
            except BaseException as e:   # noqa: BLE001 - re-raised below
                failed.append(e)
        cb = _BLOCK_CB_TYPE(_cb)
        lib().call("umpr_vgg16_set_block_callback", ctypes.cast(cb, ctypes.c_void_p), None)
        try:
            lib().call(name, *args)
        finally:
            lib().call("umpr_vgg16_set_block_callback", None, None)
        if failed:
            raise failed[0]
    else:
        lib().call(name, *args)


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr, ctypes.cast(arr, ctypes.c_void_p)


class _VGGFeatures(torch.autograd.Function):
    """Convolutional stage of torchvision.models.vgg16 configuration D (call site src/model.py:204-207,217).
    Returns (pool5 [n,25088] - a view into the activation arena -, the arena itself)."""

    @staticmethod
    def forward(ctx, images, *params):
        n = images.shape[0]
        assert tuple(images.shape[1:]) == (3, 224, 224), "VGG16 kernels take 3x224x224 images (src/dataset.py:146)"
        dev = images.device
        images = _c(images)
        params_in = params
        params = [_c(p) for p in params]
        acts = torch.empty(lib().size("umpr_vgg16_act_bytes", n) // 4, device=dev, dtype=torch.float32)
        ws, wsb = _ws(lib().size("umpr_vgg16_fwd_ws_bytes", n), dev)
        # the C side takes the full 32-pointer table; the classifier slots are not touched by this stage
        keep, parr = _ptr_array(params + params[:6])
        # no gradient will be taken through this forward (evaluate.py:8-13 runs under no_grad): the library may then use the
        # F(4x4,3x3) Winograd tile in forward too (umpr_set_conv_inference; +15 % inference throughput, predictions move 3e-6)
        infer = _MODE.infer
        if infer:
            lib().fn["umpr_set_conv_inference"](1)
        try:
            lib().call("umpr_vgg16_features_fwd", images, parr, n, acts, ws, wsb, stream_ptr())
        finally:
            if infer:
                lib().fn["umpr_set_conv_inference"](0)
        off = lib().size("umpr_vgg16_pool5_offset", n) // 4
        pool5 = acts[off:off + n * 25088].view(n, 25088)
        ctx.save_for_backward(images, acts, *params)
        ctx.param_objs = params_in
        ctx.mark_non_differentiable(acts)
        # without this autograd hands backward a zero-filled gradient for the arena output: a 3.8 GB fill kernel per step
        # (1.1 ms at batch 64, on the main stream right where the convolutional backward starts)
        ctx.set_materialize_grads(False)
        return pool5, acts

    @staticmethod
    def backward(ctx, d_pool5, _):
        images, acts, *params = ctx.saved_tensors
        n = images.shape[0]
        dev = images.device
        if d_pool5 is None:                                  # nothing downstream used the features
            d_pool5 = torch.zeros(n, 25088, device=dev)
        grads, direct = _grad_targets(ctx.param_objs)
        ws, wsb = _ws(lib().size("umpr_vgg16_features_bwd_ws_bytes", n), dev)
        keep_p, parr = _ptr_array(params + params[:6])
        keep_g, garr = _ptr_array(grads + grads[:6])
        _features_bwd_call("umpr_vgg16_features_bwd", direct, ctx.param_objs, images, parr, n, acts, _c(d_pool5), garr, ws,
                           wsb, stream_ptr())
        return (None, *_grad_returns(ctx.param_objs, grads, direct))


class _VGGFeaturesBF16(torch.autograd.Function):
    """The same convolutional stage in bf16 mixed precision (BASELINE.json configs[4]): bf16 activations / gradients in
    the library's CB8-PF layout, fp32 accumulation, fp32 master weights and weight gradients.  Returns (pool5 [n,25088]
    fp32 - a view into the compact classifier arena -, that arena)."""

    @staticmethod
    def forward(ctx, images, *params):
        n = images.shape[0]
        assert tuple(images.shape[1:]) == (3, 224, 224), "VGG16 kernels take 3x224x224 images (src/dataset.py:146)"
        dev = images.device
        images = _c(images)
        params_in = params
        params = [_c(p) for p in params]
        acts = torch.empty(lib().size("umpr_vgg16_bf16_act_bytes", n), device=dev, dtype=torch.uint8)
        cls = torch.empty(lib().size("umpr_vgg16_cls_arena_bytes", n) // 4, device=dev, dtype=torch.float32)
        ws, wsb = _ws(lib().size("umpr_vgg16_bf16_fwd_ws_bytes", n), dev)
        keep, parr = _ptr_array(params + params[:6])
        lib().call("umpr_vgg16_bf16_features_fwd", images, parr, n, acts, cls, ws, wsb, stream_ptr())
        pool5 = cls[:n * 25088].view(n, 25088)
        ctx.save_for_backward(images, acts, *params)
        ctx.param_objs = params_in
        ctx.mark_non_differentiable(cls)
        ctx.set_materialize_grads(False)
        return pool5, cls

    @staticmethod
    def backward(ctx, d_pool5, _):
        images, acts, *params = ctx.saved_tensors
        n = images.shape[0]
        dev = images.device
        if d_pool5 is None:
            d_pool5 = torch.zeros(n, 25088, device=dev)
        grads, direct = _grad_targets(ctx.param_objs)
        ws, wsb = _ws(lib().size("umpr_vgg16_bf16_bwd_ws_bytes", n), dev)
        keep_p, parr = _ptr_array(params + params[:6])
        keep_g, garr = _ptr_array(grads + grads[:6])
        _features_bwd_call("umpr_vgg16_bf16_features_bwd", direct, ctx.param_objs, images, parr, n, acts, _c(d_pool5), garr,
                           ws, wsb, stream_ptr())
        return (None, *_grad_returns(ctx.param_objs, grads, direct))


_CLS_BF16 = os.environ.get("UMPR_CLS_BF16", "1") != "0"


class _VGGClassifier(torch.autograd.Function):
    """Linear(25088,4096)-ReLU-Dropout-Linear(4096,4096)-ReLU-Dropout-Linear(4096,1000) on the pooled features."""

    @staticmethod
    def forward(ctx, pool5, acts, train, masks_in, seed, owner, compact, *params):
        """`acts`: the fp32 activation arena of _VGGFeatures, or (compact) the classifier arena of _VGGFeaturesBF16."""
        n = pool5.shape[0]
        dev = pool5.device
        ctx.param_objs = params
        ctx.owner = owner
        # compact = the bf16 path: its classifier products run on the bf16 matrix pipe too (UMPR_CLS_BF16=0: fp32 MFMA)
        ctx.suffix = ("_compact_bf16" if _CLS_BF16 else "_compact") if compact else ""
        params = [_c(p) for p in params]
        use_masks = masks_in is not None
        masks = _c(masks_in) if use_masks else torch.empty(2, n, 4096, device=dev, dtype=torch.uint8)
        out = torch.empty(n, 1000, device=dev, dtype=torch.float32)
        ws, wsb = _ws(lib().size("umpr_vgg16_fwd_ws_bytes", n), dev)
        keep, parr = _ptr_array([params[0]] * 26 + params)   # slots 26..31 = classifier
        lib().call("umpr_vgg16_classifier_fwd" + ctx.suffix, parr, n, int(train), int(use_masks), int(seed), acts, masks,
                   out, ws, wsb, stream_ptr())
        ctx.dropout = bool(train) or use_masks
        ctx.save_for_backward(acts, masks, *params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        acts, masks, *params = ctx.saved_tensors
        n = d_out.shape[0]
        dev = d_out.device
        grads, direct = _grad_targets(ctx.param_objs)
        d_pool5 = torch.empty(n, 25088, device=dev, dtype=torch.float32)
        ws, wsb = _ws(lib().size("umpr_vgg16_classifier_bwd_ws_bytes", n), dev)
        keep_p, parr = _ptr_array([params[0]] * 26 + params)
        keep_g, garr = _ptr_array([grads[0]] * 26 + grads)
        lib().call("umpr_vgg16_classifier_bwd" + ctx.suffix, parr, n, int(ctx.dropout), acts, masks, _c(d_out), garr,
                   d_pool5, ws, wsb, stream_ptr())
        out = _grad_returns(ctx.param_objs, grads, direct)
        if all(direct) and ctx.owner is not None:   # written in place: no AccumulateGrad hook will announce them
            for cb in ctx.owner.grad_callbacks:
                cb()
        return (d_pool5, None, None, None, None, None, None, *out)


# --------------------------------------------------------------------------------------------- K11-K12
class _Head(torch.autograd.Function):
    """Visual head + linear_fusion + MSE / loss_v (src/model.py:218-228, 267-277)."""

    @staticmethod
    def forward(ctx, rr, c_u, c_i, pp, pn, vgg, pos_v, neg_v, lw, lb, fw, fb, labels, rate, V, Pc):
        B = rr.shape[0]
        dev = rr.device
        f = dict(device=dev, dtype=torch.float32)
        ctx.set_materialize_grads(False)
        pred, loss, z = torch.empty(B, **f), torch.empty(3, **f), torch.empty(B, **f)
        nv = max(V, 1)
        img_emb, pm, nm = torch.empty(B, nv, **f), torch.empty(B, nv, **f), torch.empty(B, nv, **f)
        pne = torch.empty(2, nv, **f)
        rr = _c(rr)
        args = [(_c(t) if t is not None else None) for t in (c_u, c_i, pp, pn, vgg, pos_v, neg_v, lw, lb)]
        lib().call("umpr_head_fwd", rr, *args, fw, fb, labels, float(rate), B, V, Pc, pred, loss, z, img_emb, pm, nm,
                   pne, stream_ptr())
        ctx.meta = (float(rate), B, V, Pc)
        ctx.param_objs = (pos_v, neg_v, lw, lb, fw, fb)
        ctx.opt = [t is not None for t in args]
        keep = [t if t is not None else torch.empty(0, **f) for t in args]
        ctx.save_for_backward(rr, *keep, fw, labels, pred, z, img_emb, pm, nm, pne)
        terms = loss[1:]
        ctx.mark_non_differentiable(terms)
        return pred, loss[0], terms

    @staticmethod
    def backward(ctx, d_pred, d_loss, _unused):
        rr, c_u, c_i, pp, pn, vgg, pos_v, neg_v, lw, lb, fw, labels, pred, z, img_emb, pm, nm, pne = ctx.saved_tensors
        rate, B, V, Pc = ctx.meta
        dev = rr.device
        f = dict(device=dev, dtype=torch.float32)
        if d_loss is None:
            d_loss = torch.zeros((), **f)
        d_loss = d_loss.reshape(1).contiguous()
        d_pred = _c(d_pred) if d_pred is not None else None
        d_rr = torch.empty_like(rr)
        # the head's own parameters (visual head, linear_fusion): written straight into the optimiser's arena when it marked them
        # fresh (no temporaries, no AccumulateGrad add kernels - six tiny torch launches per step in round 2)
        po = ctx.param_objs
        if V > 0:
            tg, direct = _grad_targets(po)                       # (pos_v, neg_v, lw, lb, fw, fb)
            d_pos, d_neg, d_lw, d_lb, d_fw, d_fb = tg
            d_cu, d_ci, d_pp, d_pn = (torch.empty(B, V, **f) for _ in range(4))
            d_vgg = torch.empty_like(vgg)
        else:
            tg, direct = _grad_targets(po[4:])
            d_fw, d_fb = tg
            d_cu = d_ci = d_pp = d_pn = d_vgg = d_pos = d_neg = d_lw = d_lb = None
        opt = lambda t: t if (t is not None and t.numel() > 0) else None
        lib().call("umpr_head_bwd", rr, opt(c_u), opt(c_i), opt(pp), opt(pn), opt(vgg), opt(pos_v), opt(neg_v), opt(lw),
                   fw, labels, rate, B, V, Pc, pred, z, img_emb, pm, nm, pne, d_loss, d_pred, d_rr, d_cu, d_ci, d_pp,
                   d_pn, d_vgg, d_pos, d_neg, d_lw, d_lb, d_fw, d_fb, stream_ptr())
        if V > 0:
            r = _grad_returns(po, tg, direct)
        else:
            r = [None] * 4 + _grad_returns(po[4:], tg, direct)
        return d_rr, d_cu, d_ci, d_pp, d_pn, d_vgg, r[0], r[1], r[2], r[3], r[4], r[5], None, None, None, None


# --------------------------------------------------------------------------------------------- module tree
class ImprovedRnn(nn.Module):
    """Parameter holder with the reference's name ``gru.module.*`` (src/model.py:6-10)."""

    def __init__(self, module, *args, **kwargs):
        assert module is nn.GRU, "the MI355X path implements the GRU the reference instantiates"
        super().__init__()
        self.module = module(*args, **kwargs)
        assert 1 <= self.module.hidden_size <= H and self.module.bidirectional and self.module.num_layers == 1, \
            f"gru_size must be 1..{H} (config.py:34 uses 64: the kernels' width; smaller sizes are embedded in it), bidirectional, one layer"

    def weights(self):
        m = self.module
        return (m.weight_ih_l0, m.weight_hh_l0, m.bias_ih_l0, m.bias_hh_l0, m.weight_ih_l0_reverse,
                m.weight_hh_l0_reverse, m.bias_ih_l0_reverse, m.bias_hh_l0_reverse)

    def forward(self, ids, lengths_dev, order_dev, emb, split=0):
        """ids [N,L] int64 on device; lengths/order int32 on device -> [N, L, 128] (two tensors when split > 0)."""
        return _EmbedGru.apply(ids, lengths_dev, order_dev, emb, split, *self.weights())


class RNet(nn.Module):
    def __init__(self, gru_in, gru_out):
        super().__init__()
        self.gru = ImprovedRnn(nn.GRU, input_size=gru_in, hidden_size=gru_out, batch_first=True, bidirectional=True)
        self.M = nn.Parameter(torch.randn(2 * gru_out, 2 * gru_out))


class SNet(nn.Module):
    def __init__(self, self_atte_size, repr_size):
        super().__init__()
        assert 1 <= self_atte_size <= AT and repr_size <= D, \
            f"self_atte_size must be 1..{AT} and gru_size 1..{H} (the kernels' widths; smaller sizes are embedded in them)"
        self.Ms = nn.Parameter(torch.randn(self_atte_size, repr_size))
        self.Ws = nn.Parameter(torch.randn(1, self_atte_size))


class CNet(nn.Module):
    def __init__(self, gru_in, gru_out, k_count, k_size, view_size, threshold=0.35):
        super().__init__()
        # any Conv1d width: an even k_size yields L-1 positions in the reference (src/model.py:93, padding (k-1)//2), which the
        # C-Net kernels reproduce (umpr_cnet_head_fwd: the maximum runs over the valid positions)
        assert 1 <= k_size <= 8, f"kernel_size={k_size}: the C-Net kernels take Conv1d widths 1..8 (config.py:37 uses 3)"
        self.threshold = threshold
        self.gru = ImprovedRnn(nn.GRU, input_size=gru_in, hidden_size=gru_out, batch_first=True, bidirectional=True)
        self.cnn = nn.Sequential(nn.Conv1d(2 * gru_out, k_count, k_size, padding=(k_size - 1) // 2), nn.ReLU())
        self.linear = nn.Sequential(nn.Linear(k_count, view_size), nn.Sigmoid())


class SSNet(nn.Module):
    def __init__(self, input_size):
        super().__init__()
        self.linear = nn.Sequential(nn.Linear(input_size, 1), nn.Sigmoid())


class ReviewNet(nn.Module):
    def __init__(self, emb_size, gru_size, atte_size):
        super().__init__()
        self.r_net = RNet(emb_size, gru_size)
        self.s_net_u = SNet(atte_size, gru_size * 2)
        self.s_net_i = SNet(atte_size, gru_size * 2)
        self.linear_u = nn.Linear(gru_size * 4, gru_size * 2, bias=False)
        self.linear_i = nn.Linear(gru_size * 4, gru_size * 2, bias=False)


class ControlNet(nn.Module):
    def __init__(self, emb_size, gru_size, k_count, k_size, view_size, threshold, atte_size):
        super().__init__()
        self.c_net = CNet(emb_size, gru_size, k_count, k_size, view_size, threshold)
        self.s_net = SNet(atte_size, repr_size=gru_size * 2)
        self.ss_net = SSNet(input_size=gru_size * 2)


_VGG_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")


class VGG16(nn.Module):
    """Layer list of torchvision's vgg16 (cfg "D") so that state_dict keys match ``features.N`` / ``classifier.N``."""

    def __init__(self, num_classes=1000, dtype="fp32"):
        super().__init__()
        assert dtype in ("fp32", "bf16"), dtype
        self.compute_dtype = dtype   # "bf16": conv stack and classifier products on bf16 MFMA, fp32 accumulation
        layers, cin = [], 3
        for v in _VGG_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(),
                                        nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, num_classes))
        for m in self.modules():  # torchvision's non-pretrained initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)
        self.dropout_masks = None  # optional injected keep-masks uint8 [2][n][4096] (parity tests)
        self._calls = 0
        self.grad_callbacks = []   # called once the classifier gradients have been written in place (GradReducer)
        for p in self.parameters():
            p._umpr_direct = True  # one backward node writes each of these exactly once per step

    def param_list(self):
        ps = []
        for m in self.features:
            if isinstance(m, nn.Conv2d):
                ps += [m.weight, m.bias]
        for m in self.classifier:
            if isinstance(m, nn.Linear):
                ps += [m.weight, m.bias]
        return ps

    def forward(self, images):
        self._calls += 1
        seed = (torch.initial_seed() * 1000003 + self._calls) & 0x7FFFFFFFFFFFFFFF
        ps = self.param_list()
        bf16 = self.compute_dtype == "bf16"
        _MODE.infer = not torch.is_grad_enabled()
        try:
            pool5, acts = (_VGGFeaturesBF16 if bf16 else _VGGFeatures).apply(images, *ps[:26])
        finally:
            _MODE.infer = False
        return _VGGClassifier.apply(pool5, acts, self.training, self.dropout_masks, seed, self, bf16, *ps[26:])


class VisualNet(nn.Module):
    def __init__(self, view_size, vgg_out=1000, vgg_weights=None, dtype="fp32"):
        super().__init__()
        # The reference asks torchvision for ImageNet weights (pretrained=True, src/model.py:205): a network fetch.
        # Offline the stack is randomly initialised; pass vgg_weights=<local vgg16 state_dict .pth> to load a file.
        self.vgg16 = nn.Sequential(VGG16(vgg_out, dtype=dtype))
        if vgg_weights:
            self.vgg16[0].load_state_dict(torch.load(vgg_weights, map_location="cpu", weights_only=True))
        self.pos_v_emb = nn.Parameter(torch.randn(view_size, vgg_out))
        self.neg_v_emb = nn.Parameter(torch.randn(view_size, vgg_out))
        self.linear = nn.Linear(vgg_out, 1)


class UMPR(nn.Module):
    def __init__(self, config, word_emb):
        super().__init__()
        self.review_net_only = config.review_net_only
        self.loss_v_rate = config.loss_v_rate
        # not in the reference: `--dtype bf16` = mixed precision (BASELINE.json configs[4]) - bf16 MFMA for the VGG conv
        # stack and the co-attention scores, fp32 accumulation, fp32 master weights / GRU gates / softmax / Adam
        self.compute_dtype = str(getattr(config, "dtype", "fp32"))
        assert self.compute_dtype in ("fp32", "bf16"), f"dtype must be fp32 or bf16, got {self.compute_dtype}"
        # one wave handles one sentence in the S-Net kernels, up to four positions per lane: at most 256 tokens per sentence
        # (config.py:29 uses 20; review_level='review', src/dataset.py:24, makes whole reviews the "sentences")
        assert int(getattr(config, "max_sent_length", 20)) <= 256, "max_sent_length > 256 is not supported by the S-Net kernels"
        self.gru_size, self.atte_size = int(config.gru_size), int(config.self_atte_size)
        assert 1 <= self.gru_size <= H and 1 <= self.atte_size <= AT, \
            f"gru_size={self.gru_size} / self_atte_size={self.atte_size}: the kernels are {H} / {AT} wide; smaller sizes run embedded " \
            f"in them (exactly), larger ones are not supported"
        self._embedded = self.gru_size != H or self.atte_size != AT
        self._pad_maps = None
        self._static_index = None
        self.views = list(getattr(config, "views", []))
        self.embedding = nn.Embedding.from_pretrained(torch.Tensor(word_emb))
        E = self.embedding.embedding_dim
        self.review_net = ReviewNet(E, config.gru_size, config.self_atte_size)
        if config.review_net_only:
            self.linear_fusion = nn.Sequential(nn.Linear(config.gru_size * 2, 1), nn.ReLU())
        else:
            view_size = len(config.views)
            self.control_net = ControlNet(E, config.gru_size, config.kernel_count, config.kernel_size, view_size,
                                          config.threshold, config.self_atte_size)
            self.visual_net = VisualNet(view_size, vgg_weights=getattr(config, "vgg_weights", None),
                                        dtype=self.compute_dtype)
            self.linear_fusion = nn.Sequential(nn.Linear(config.gru_size * 2 + view_size + view_size, 1), nn.ReLU())
        self.last_loss_terms = None
        # Parameters whose backward node can write the gradient straight into the optimiser's arena (_grad_targets):
        # everything in the review / control nets (GRU weights accumulate in place across their uses) and, set by VGG16
        # itself, the VGG stack; the head's own parameters (visual head, linear_fusion) by the head node.
        # (embedded sizes: the backward nodes see zero-padded copies, and the real parameters receive their gradients through
        # autograd's accumulation - they are zeroed and accumulated like any other torch parameter)
        for mod in (self.review_net, getattr(self, "control_net", None), self.linear_fusion):
            if mod is not None:
                for p in mod.parameters():
                    p._umpr_direct = not self._embedded
        if not config.review_net_only:
            vn = self.visual_net
            for p in (vn.pos_v_emb, vn.neg_v_emb, vn.linear.weight, vn.linear.bias):
                p._umpr_direct = True      # written by the head's backward node (_Head), once per step

    @staticmethod
    def _host_perm(lengths, device):
        """lengths stay on the host (src/model.py:18 ``lengths.cpu()``): the descending, NON-stable torch.sort that
        pack_padded_sequence runs defines the sentence permutation (SURVEY.md header fact 1)."""
        flat = lengths.reshape(-1).cpu()
        _, sorted_indices = torch.sort(flat, descending=True)
        both = torch.stack([flat.to(torch.int32), sorted_indices.to(torch.int32)])
        both = both.to(device, non_blocking=True)
        return both[0], both[1]

    @staticmethod
    def _side_stream(device, which=0):
        """High-priority side streams of the text path: its many small latency-bound kernels should not queue behind the
        big convolution grids.  Stream 0 carries the ReviewNet, stream 1 the ControlNet (independent until the head)."""
        st = _SIDE_STREAMS.get((device, which))
        if st is None:
            st = _SIDE_STREAMS[(device, which)] = torch.cuda.Stream(device, priority=int(os.environ.get("UMPR_TEXT_PRIO", "-1")))
        return st

    @staticmethod
    def _pair(user_reviews, item_reviews, lu, ou, li, oi):
        """User and item reviews as ONE batch of 2N sequences for the GRUs they share (src/model.py:45-46 and :183-184 call
        the same module on both): the recurrent kernels are bound by the latency of their <= 20 dependent time steps, not by
        work, so one launch over 2N sequences costs what a launch over N does, and the input-projection GEMM doubles its M.
        Each half keeps its own permutation (the reference's double un-sort is per call): item rows and their destination
        rows are offset by N.  (Stage-level form with device-side concatenations: kept for callers that hold device index
        tensors; UMPR.forward builds the pair on the host, _index_upload.)"""
        B, S, L = user_reviews.shape
        N = B * S
        ids = torch.cat([user_reviews.view(N, L), item_reviews.view(N, L)])
        return ids, torch.cat([lu, li]), torch.cat([ou, oi + N]), N

    _INDEX_RING = {}     # device -> [pinned host buffers, events, position]: staging for the per-step index upload

    @classmethod
    def _index_upload(cls, u_lengths, i_lengths, ui_lengths, device):
        """lengths stay on the host (src/model.py:18 ``lengths.cpu()``): the descending, NON-stable torch.sort that
        pack_padded_sequence runs defines each review tensor's sentence permutation (SURVEY.md header fact 1).  Everything the
        kernels need of it - lengths and sorted indices of the user+item pair (item half offset by N) and of the ui reviews -
        travels to the device as ONE int32 tensor per step, from a small ring of pinned buffers (a slot is reused 8 steps later;
        its event is only waited for if the GPU is that far behind)."""
        parts = []
        flats = []
        for ln in (u_lengths, i_lengths):
            flat = ln.reshape(-1).cpu()
            _, si = torch.sort(flat, descending=True)
            flats.append((flat, si))
        N = flats[0][0].numel()
        parts += [flats[0][0], flats[1][0], flats[0][1], flats[1][1] + N]
        n_ui = 0
        if ui_lengths is not None:
            flat = ui_lengths.reshape(-1).cpu()
            _, si = torch.sort(flat, descending=True)
            n_ui = flat.numel()
            parts += [flat, si]
        total = 4 * N + 2 * n_ui
        if device is None:           # host part only (umpr_amd/graphs.py copies it into its static buffer)
            return torch.cat(parts).to(torch.int32), N, n_ui
        ring = cls._INDEX_RING.get(device)
        if ring is None or ring[0][0].numel() < total:
            ring = cls._INDEX_RING[device] = [[torch.empty(max(total, 4096), dtype=torch.int32).pin_memory() for _ in range(8)],
                                              [None] * 8, 0]
        k = ring[2] % 8
        ring[2] += 1
        if ring[1][k] is not None:
            ring[1][k].synchronize()
        host = ring[0][k][:total]
        host.copy_(torch.cat(parts))        # int64 -> int32 on the host
        devt = host.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        ring[1][k] = ev
        lens_pair, ord_pair = devt[:2 * N], devt[2 * N:4 * N]
        lens_ui, ord_ui = (devt[4 * N:4 * N + n_ui], devt[4 * N + n_ui:]) if n_ui else (None, None)
        return devt, lens_pair, ord_pair, lens_ui, ord_ui

    def _maps(self, device):
        if self._pad_maps is None or self._pad_maps.hid.device != device:
            self._pad_maps = _PadMaps(self.gru_size, self.atte_size, 0 if self.review_net_only else len(self.views), device)
        return self._pad_maps

    def _gru_params(self, gru, device):
        w = gru.weights()
        if not self._embedded:
            return w
        m = self._maps(device)
        E = w[0].shape[1]
        out = []
        for k in range(2):   # forward direction, then "_reverse"
            wih, whh, bih, bhh = w[4 * k: 4 * k + 4]
            out += [_ZeroPad.apply(wih, (3 * H, E), m.gate, None), _ZeroPad.apply(whh, (3 * H, H), m.gate, m.hid),
                    _ZeroPad.apply(bih, (3 * H,), m.gate, None), _ZeroPad.apply(bhh, (3 * H,), m.gate, None)]
        return tuple(out)

    def _snet_params(self, sn, device):
        if not self._embedded:
            return sn.Ms, sn.Ws
        m = self._maps(device)
        return _ZeroPad.apply(sn.Ms, (AT, D), m.att, m.rep), _ZeroPad.apply(sn.Ws, (1, AT), None, m.att)

    def _review_params(self, device=None):
        rn = self.review_net
        lin = (rn.linear_u.weight, rn.linear_i.weight)
        Mx = rn.r_net.M
        if self._embedded:
            m = self._maps(device)
            Mx = _ZeroPad.apply(Mx, (D, D), m.rep, m.rep)
            lin = tuple(_ZeroPad.apply(wt, (D, 2 * D), m.rep, m.rep2) for wt in lin)
        return (*self._gru_params(rn.r_net.gru, device), Mx, *self._snet_params(rn.s_net_u, device),
                *self._snet_params(rn.s_net_i, device), *lin)

    def _control_params(self, device=None):
        cn = self.control_net
        Wc, ssW = cn.c_net.cnn[0].weight, cn.ss_net.linear[0].weight
        if self._embedded:
            m = self._maps(device)
            Wc = _ZeroPad.apply(Wc, (Wc.shape[0], D, Wc.shape[2]), None, m.rep)
            ssW = _ZeroPad.apply(ssW, (1, D), None, m.rep)
        return (*self._gru_params(cn.c_net.gru, device), Wc, cn.c_net.cnn[0].bias, cn.c_net.linear[0].weight,
                cn.c_net.linear[0].bias, *self._snet_params(cn.s_net, device), ssW, cn.ss_net.linear[0].bias)

    def _fusion_weight(self, device):
        fw = self.linear_fusion[0].weight
        if not self._embedded:
            return fw
        m = self._maps(device)
        return _ZeroPad.apply(fw, (1, D + (0 if self.review_net_only else 2 * len(self.views))), None, m.fus)

    def forward(self, user_reviews, item_reviews, ui_reviews, u_lengths, i_lengths, ui_lengths, photos, labels):
        _MODE.b16 = self.compute_dtype == "bf16" and _TEXT_BF16
        try:
            return self._forward(user_reviews, item_reviews, ui_reviews, u_lengths, i_lengths, ui_lengths, photos, labels)
        finally:
            _MODE.b16 = False

    def _forward(self, user_reviews, item_reviews, ui_reviews, u_lengths, i_lengths, ui_lengths, photos, labels):
        device = self.embedding.weight.device
        if device.type != "cuda":
            raise RuntimeError("umpr_amd.UMPR runs on an MI355X only (no CPU fallback): move the module to a cuda device")
        # pinned host batches (DataLoader(pin_memory=True)) upload asynchronously; the kernels follow on the same stream
        user_reviews, item_reviews, ui_reviews = [_c(v.to(device, non_blocking=True))
                                                  for v in (user_reviews, item_reviews, ui_reviews)]
        photos, labels = [v.to(device, non_blocking=True) for v in (photos, labels)]
        labels = _c(labels.float())
        emb = self.embedding.weight
        B, S, L = user_reviews.shape
        _, S_ui, L_ui = ui_reviews.shape
        N = B * S
        b16_scores = self.compute_dtype == "bf16"
        b16_gemm = b16_scores and _TEXT_BF16
        fus = self.linear_fusion[0]
        main = torch.cuda.current_stream(device)
        full = not self.review_net_only
        side = self._side_stream(device, 0) if (TEXT_STREAM and full) else None
        if side is not None:
            side.wait_stream(main)
        # The text path (many small, latency-bound kernels: GRUs, co-attention, heads) runs on a side stream beside the
        # VGG stack (MFMA-bound) and joins it at the head; autograd replays the same split in backward.
        _TEXT_TLS.on_side = side is not None
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            if self._static_index is not None:      # a captured step (umpr_amd/graphs.py): the caller refreshed this buffer
                idx, lens, order, lens_ui, ord_ui = self._static_index
            else:
                idx, lens, order, lens_ui, ord_ui = self._index_upload(u_lengths, i_lengths, ui_lengths if full else None, device)
            if (item_reviews.data_ptr() == user_reviews.data_ptr() + N * L * 8
                    and item_reviews.untyped_storage().data_ptr() == user_reviews.untyped_storage().data_ptr()):
                # the caller keeps the two back to back in one allocation (umpr_amd/graphs.py): that IS the [2N][L] tensor
                ids_pair = user_reviews.as_strided((2 * N, L), (L, 1))
            else:
                ids_pair = torch.empty(2 * N, L, device=device, dtype=torch.int64)
                lib().call("umpr_concat_ids", user_reviews, item_reviews, N * L, ids_pair, stream_ptr())
            rr = _ReviewNetF.apply(ids_pair, lens, order, emb, (B, S, L), b16_gemm, b16_scores, *self._review_params(device))
            if full:
                cu, ci, pp, pn = _ControlNetF.apply(ui_reviews.view(B * S_ui, L_ui), ids_pair, lens_ui, ord_ui, lens, order, emb,
                                                    (B, S_ui, L_ui, S, L), self.control_net.c_net.threshold, b16_gemm,
                                                    *self._control_params(device))
        _TEXT_TLS.on_side = False
        fus_w = self._fusion_weight(device)
        if not full:
            pred, loss, terms = _Head.apply(rr, None, None, None, None, None, None, None, None, None, fus_w,
                                            fus.bias, labels, 0.0, 0, 0)
            self.last_loss_terms = terms
            return pred, loss
        if side is not None:
            for t in (user_reviews, item_reviews, ui_reviews, idx, ids_pair):
                t.record_stream(side)
        vn = self.visual_net
        V, Pc = photos.shape[1], photos.shape[2]
        vgg = vn.vgg16[0](photos.reshape(B * V * Pc, *photos.shape[3:]).float())
        if side is not None:
            main.wait_stream(side)
            for t in (rr, cu, ci, pp, pn):
                t.record_stream(main)
        pred, loss, terms = _Head.apply(rr, cu, ci, pp, pn, vgg, vn.pos_v_emb, vn.neg_v_emb, vn.linear.weight,
                                        vn.linear.bias, fus_w, fus.bias, labels, self.loss_v_rate, V, Pc)
        self.last_loss_terms = terms
        return pred, loss
