"""The decoder layer's self-attention in TRAINING (pdvc/deformable_transformer.py:263-270:
``self.self_attn(q, k, tgt, key_padding_mask=~query_mask)`` with q = k = tgt + query_pos, an nn.MultiheadAttention with dropout on
the attention weights) on the hand-written kernels:

    in-projection   ONE gvl_linear_f16x3_f32 launch over in_proj_weight: columns [0, 2C) multiply tgt + query_pos (the addend is
                    applied in the kernel's load path), columns [2C, 3C) multiply tgt; the epilogue leaves the row maxima of the
                    q | k and of the v columns
    attention core  gvl_mha_train_forward_f32 / _backward_f32 (scores, softmax, dropout stay in registers)
    out-projection  gvl_amd.linear.train_linear

PyTorch runs this as 3 projection GEMMs + 6 elementwise launches + bmm / softmax / dropout / bmm (~250 us forward, ~420 us
backward per layer at B = 16, Q = 300: profiles/r05_train_timeline_before.txt).  The module keeps its parameters and state-dict
keys (``in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias``); this file only replaces what its forward launches.
``GVL_TRAIN_MHA=torch`` keeps nn.MultiheadAttention's own forward (A/B switch)."""
import os

import torch

from . import _lib
from . import layers as L
from . import linear as GL
from . import train_layers as TL
from . import MultiScaleDeformableAttention as MSDA
from .train_planes import Operand

_ENABLED = os.environ.get("GVL_TRAIN_MHA", "") != "torch"


def enabled(on=None):
    global _ENABLED
    if on is not None:
        _ENABLED = bool(on)
    return _ENABLED


class _Sub:
    """a K-stage range of operand planes (planes are K-stage-major, so a range of the contraction is a contiguous piece)"""
    __slots__ = ("hi", "lo", "scale")

    def __init__(self, planes, stage0, stage1):
        self.hi, self.lo, self.scale = planes.hi[stage0:stage1], planes.lo[stage0:stage1], planes.scale


@MSDA.keeps_products
class _InProj(torch.autograd.Function):
    """qkv = [ (x + pos) W_qk^T | x W_v^T ] + b   ->  (qkv (R, 3C), row maxima of the q | k columns, of the v columns)"""

    @staticmethod
    def forward(ctx, x, pos, xq, weight, bias, defer=False):
        ctx.defer, ctx.params = bool(defer), (weight, bias)
        # x (B, Q, C) contiguous; pos (Q, C) (row stride arbitrary); xq = x + pos as a tensor (the weight gradient's operand)
        B, Q, C = x.shape
        R = B * Q
        x2 = x.reshape(R, C)
        op, op_t = GL._operands((weight,), (bias,))
        am_v = GL._row_amax(x2, x)
        am_qk = GL._row_amax(xq.reshape(R, C), xq)
        qkv = torch.empty(R, 3 * C, device=x.device, dtype=torch.float32)
        am_out = TL.step_zeros(2 * R, x.device).view(2, R)
        L.linear(x2, op, [L.seg(0, qkv[:, :2 * C], am_qk, amax_out=am_out[0], addend=True),
                          L.seg(2 * C, qkv[:, 2 * C:], am_v, amax_out=am_out[1])], a2=pos)
        ctx.save_for_backward(x2, xq.reshape(R, C), am_v, am_qk, weight)
        ctx.op_t, ctx.shape = op_t, (B, Q, C)
        ctx.mark_non_differentiable(am_out)
        ctx.set_materialize_grads(False)          # (the non-differentiable by-product would otherwise get a zeros() launch in backward)
        return qkv, am_out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dqkv, _dam):
        x2, xq2, am_v, am_qk, weight = ctx.saved_tensors
        B, Q, C = ctx.shape
        R = B * Q
        op_t = ctx.op_t
        hit = getattr(dqkv, "_gvl_amax_qk_v", None)                       # (row maxima left by the attention backward kernels)
        dqkv = dqkv.contiguous()
        g_qk, g_v = dqkv[:, :2 * C], dqkv[:, 2 * C:]
        if hit is not None and hit[1] == dqkv._version and hit[0].shape == (2, R):
            am_gqk, am_gv = hit[0][0], hit[0][1]
        else:
            am_gqk, am_gv = L.row_absmax(g_qk)[0], L.row_absmax(g_v)[0]
        dxq, dxv, gw, gb = _in_proj_gradients(ctx, g_qk, g_v, am_gqk, am_gv, x2, xq2, am_v, am_qk, C)
        return dxv.view(B, Q, C), None, dxq.view(B, Q, C), gw, gb, None


def _in_proj_gradients(ctx, g_qk, g_v, am_gqk, am_gv, x2, xq2, am_v, am_qk, C):
    """(d (x + pos), d x [v path], d in_proj_weight, d in_proj_bias) from the gradients of the q | k and the v columns"""
    op_t, R, dev = ctx.op_t, g_qk.shape[0], g_qk.device
    # dx of the two operand groups: the transposed planes cut at contraction stage 2C / 32
    cut = 2 * C // 32
    dxq = torch.empty(R, C, device=dev, dtype=torch.float32)
    dxv = torch.empty(R, C, device=dev, dtype=torch.float32)
    L.linear(g_qk, Operand(_Sub(op_t.planes, 0, cut), C, 2 * C, None), [L.seg(0, dxq, am_gqk)])
    L.linear(g_v, Operand(_Sub(op_t.planes, cut, 3 * C // 32), C, C, None), [L.seg(0, dxv, am_gv)])
    gw = torch.empty(3 * C, C, device=dev, dtype=torch.float32)
    gb = torch.empty(3 * C, device=dev, dtype=torch.float32)
    q = GL._WGRAD_QUEUE
    if ctx.defer and q is not None and all(p_.grad is None for p_ in ctx.params):
        # the two halves as ONE queued problem each would return two tensors per parameter; instead: one (3C, C) gradient whose
        # row blocks are two entries of the queue (filled when the group runs)
        gwd, gbd = gw.detach(), gb.detach()
        if any(id(p_) in q.seen for p_ in ctx.params):
            q.flush()
            MSDA.wgrad(g_qk, xq2, am_gqk, am_qk, grad_w=gwd[:2 * C], grad_b=gbd[:2 * C])
            MSDA.wgrad(g_v, x2, am_gv, am_v, grad_w=gwd[2 * C:], grad_b=gbd[2 * C:])
        else:
            q.seen.update(id(p_) for p_ in ctx.params)
            if len(q.items) + 2 > MSDA.wgrad_group_max():
                q.flush()
            q.items.append((g_qk, xq2, am_gqk, am_qk, gwd[:2 * C], gbd[:2 * C]))
            q.items.append((g_v, x2, am_gv, am_v, gwd[2 * C:], gbd[2 * C:]))
        return dxq, dxv, gw, gb
    MSDA.wgrad(g_qk, xq2, am_gqk, am_qk, grad_w=gw[:2 * C], grad_b=gb[:2 * C])
    MSDA.wgrad(g_v, x2, am_gv, am_v, grad_w=gw[2 * C:], grad_b=gb[2 * C:])
    return dxq, dxv, gw, gb


@MSDA.keeps_products
class _InProjShared(torch.autograd.Function):
    """_InProj for the FIRST decoder layer under the 'queries' input: every video's rows are the same Q rows of the query embedding
    (deformable_transformer.py:128-135), so q, k, v are computed for those Q rows once and copied to the B videos (the attention
    core and what follows differ per video: dropout), and the backward sums dqkv over the videos first -- the products run on Q rows
    instead of B Q.  x_rows, pos_rows (Q, C): the embedding's two blocks; -> (qkv (B Q, 3C), row maxima (2, B Q))"""

    @staticmethod
    def forward(ctx, x_rows, pos_rows, weight, bias, B, defer=False):
        ctx.defer, ctx.params = bool(defer), (weight, bias)
        Q, C = x_rows.shape
        x2 = x_rows.contiguous()
        p2 = pos_rows if pos_rows.stride(1) == 1 and pos_rows.stride(0) % 4 == 0 and pos_rows.data_ptr() % 16 == 0 \
            else pos_rows.contiguous()
        xq2 = x2 + p2
        op, op_t = GL._operands((weight,), (bias,))
        am_v, am_qk = L.row_absmax(x2, p2)
        q1 = torch.empty(Q, 3 * C, device=x2.device, dtype=torch.float32)
        am1 = TL.step_zeros(2 * Q, x2.device).view(2, Q)
        L.linear(x2, op, [L.seg(0, q1[:, :2 * C], am_qk, amax_out=am1[0], addend=True),
                          L.seg(2 * C, q1[:, 2 * C:], am_v, amax_out=am1[1])], a2=p2)
        qkv = q1.unsqueeze(0).expand(B, Q, 3 * C).reshape(B * Q, 3 * C)
        am_out = am1.unsqueeze(1).expand(2, B, Q).reshape(2, B * Q)
        ctx.save_for_backward(x2, xq2, am_v, am_qk, weight)
        ctx.op_t, ctx.shape = op_t, (B, Q, C)
        ctx.mark_non_differentiable(am_out)
        ctx.set_materialize_grads(False)
        return qkv, am_out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dqkv, _dam):
        import ctypes
        x2, xq2, am_v, am_qk, weight = ctx.saved_tensors
        B, Q, C = ctx.shape
        dqkv = dqkv.contiguous()
        d1 = torch.empty(Q, 3 * C, device=dqkv.device, dtype=torch.float32)
        with torch.cuda.device(dqkv.device):
            rc = _lib.lib().gvl_batch_sum_f32((ctypes.c_void_p * 1)(dqkv.data_ptr()), 1, B, Q, 3 * C, d1.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "batch_sum")
        g_qk, g_v = d1[:, :2 * C], d1[:, 2 * C:]
        am_gqk, am_gv = L.row_absmax(g_qk)[0], L.row_absmax(g_v)[0]
        dxq, dxv, gw, gb = _in_proj_gradients(ctx, g_qk, g_v, am_gqk, am_gv, x2, xq2, am_v, am_qk, C)
        return dxv + dxq, dxq, gw, gb, None, None


class _Core(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, am, keep, B, Q, H, p, seed, step):
        R = B * Q
        out = torch.empty(R, H * 64, device=qkv.device, dtype=torch.float32)
        lse = torch.empty(B, H, Q, device=qkv.device, dtype=torch.float32)
        am_o = TL.step_zeros(R, qkv.device)                     # max |out row|: the row scale of the out-projection behind it
        used = TL.step_snapshot(qkv.device) if step is not None else None   # the step whose masks this forward drew
        with torch.cuda.device(qkv.device):
            rc = _lib.lib().gvl_mha_train_forward_f32(
                qkv.data_ptr(), qkv.stride(0), keep.data_ptr() if keep is not None else None, am[0].data_ptr(), am[1].data_ptr(),
                B, Q, H, float(p), int(seed), used.data_ptr() if used is not None else None, out.data_ptr(), lse.data_ptr(),
                am_o.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mha_train_forward")
        ctx.save_for_backward(qkv, am, keep if keep is not None else qkv.new_empty(0), out, lse,
                              used if used is not None else qkv.new_empty(0))
        ctx.cfg = (B, Q, H, float(p), int(seed), keep is not None, used is not None)
        ctx.mark_non_differentiable(am_o)
        ctx.set_materialize_grads(False)
        return out, am_o

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout, _dam=None):
        qkv, am, keep, out, lse, used = ctx.saved_tensors
        B, Q, H, p, seed, has_keep, has_step = ctx.cfg
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(B, H, Q, device=qkv.device, dtype=torch.float32)
        am_g = torch.empty(B * Q, device=qkv.device, dtype=torch.float32)
        am_d = TL.step_zeros(2 * B * Q, qkv.device).view(2, B * Q)       # max |row| of [dq | dk] and of dv: left by the kernels
        with torch.cuda.device(qkv.device):
            rc = _lib.lib().gvl_mha_train_backward_amax_f32(
                qkv.data_ptr(), qkv.stride(0), keep.data_ptr() if has_keep else None, am[0].data_ptr(), am[1].data_ptr(), B, Q, H,
                p, seed, used.data_ptr() if has_step else None, out.data_ptr(), lse.data_ptr(), dout.data_ptr(), delta.data_ptr(),
                am_g.data_ptr(), dqkv.data_ptr(), am_d[0].data_ptr(), am_d[1].data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mha_train_backward")
        dqkv._gvl_amax_qk_v = (am_d, dqkv._version)                       # (for the in-projection's backward: versioned like tag_amax)
        return dqkv, None, None, None, None, None, None, None, None


def eligible(mha, tgt, query_pos):
    C = tgt.shape[-1]
    return (_ENABLED and GL.train_linear_enabled() and isinstance(mha, torch.nn.MultiheadAttention) and mha._qkv_same_embed_dim
            and mha.in_proj_bias is not None and mha.bias_k is None and not mha.add_zero_attn and not mha.batch_first
            and tgt.is_cuda and tgt.dtype == torch.float32 and torch.is_grad_enabled() and not torch.is_autocast_enabled()
            and tgt.dim() == 3 and C == mha.embed_dim and C // mha.num_heads == 64 and C % 64 == 0
            and tgt.shape[0] * tgt.shape[1] >= GL.MIN_TRAIN_ROWS and tgt.shape[1] <= 4096
            and tgt.shape[0] * mha.num_heads * tgt.shape[1] * tgt.shape[1] < 2 ** 32
            and query_pos is not None and query_pos.dim() == 3 and query_pos.shape == tgt.shape and query_pos.stride(0) == 0
            and query_pos.stride(2) == 1 and query_pos.stride(1) % 4 == 0 and query_pos.data_ptr() % 16 == 0
            and query_pos.dtype == torch.float32)


def self_attention(mha, tgt, query_pos, query_mask, tgt_q=None):
    """nn.MultiheadAttention(q = k = tgt + query_pos, v = tgt, key_padding_mask = ~query_mask)[0] for batch-major tgt (B, Q, C)
    -> (B, Q, C); callers check `eligible` first.  query_pos: the batch-expanded (stride 0) query embedding."""
    B, Q, C = tgt.shape
    defer = bool(mha.__dict__.get("_gvl_defer_wgrad", False))
    rows, pos_rows = getattr(tgt, "_gvl_rows", None), getattr(query_pos, "_gvl_rows", None)
    if (rows is not None and pos_rows is not None and tgt.stride(0) == 0 and (tgt_q is None or tgt_q is tgt) and tuple(rows.shape) == (Q, C)
            and tuple(pos_rows.shape) == (Q, C) and os.environ.get("GVL_INPROJ_SHARED", "1") != "0"):
        # the first layer under the 'queries' input: the same Q rows for every video (deformable_transformer.py:128-135)
        qkv, am = _InProjShared.apply(rows, pos_rows, mha.in_proj_weight, mha.in_proj_bias, B, defer)
    else:
        x = tgt.contiguous()
        # (tgt_q: another handle of tgt for the query -- residual_dropout_norm(fan=...); the sum carries its row maxima when norm3
        #  left them)
        xq = TL.add_pos(tgt_q if tgt_q is not None else tgt, query_pos)
        L_ = L
        if L_.amax_of(x, B * Q) is None and L_.amax_of(tgt, B * Q) is not None:
            L_.tag_amax(x, L_.amax_of(tgt, B * Q))
        qkv, am = _InProj.apply(x, query_pos[0], xq, mha.in_proj_weight, mha.in_proj_bias, defer)
    keep = None
    if query_mask is not None:                          # (a bool mask's bytes ARE 0 / 1: no cast launch)
        keep = query_mask.contiguous().view(torch.uint8) if query_mask.dtype == torch.bool else query_mask.to(torch.uint8).contiguous()
    p = mha.dropout if mha.training else 0.0
    site = mha.__dict__.get("_gvl_site_drop")
    if site is None:
        site = mha.__dict__["_gvl_site_drop"] = torch.nn.Dropout(mha.dropout)
    o, am_o = _Core.apply(qkv, am, keep, B, Q, mha.num_heads, p, TL._site_seed(site),
                          TL.step_counter(tgt.device) if p > 0 else None)
    L.tag_amax(o, am_o)
    ob = mha.out_proj.bias
    if GL.train_linear_eligible(o, (mha.out_proj.weight,), (ob,)):
        return GL.train_linear(o, (mha.out_proj.weight,), (ob,), defer).view(B, Q, C)
    return torch.nn.functional.linear(o, mha.out_proj.weight, ob).view(B, Q, C)
