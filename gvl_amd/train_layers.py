"""The residual chains of the encoder / decoder layers in TRAINING -- ``norm(x + dropout(sub))`` of
pdvc/deformable_transformer.py:189-199 (encoder norm1 / norm2) and :266-280 (decoder norm2 / norm1 / norm3) -- as one
hand-written forward and one backward kernel (gvl_train_layers.hip: gvl_residual_dropout_layer_norm_{forward,backward}_f32)
instead of PyTorch's fused_dropout + add + layer_norm and their five backward launches.

Dropout masks come from a counter-based hash of (seed, step, element index): ``seed`` is derived on the host from
``torch.initial_seed()`` and the Dropout module's site number, ``step`` is a DEVICE counter that
``DeformableTransformer.forward_encoder`` advances once per training forward (one tiny kernel), so a step replayed from a hipGraph
draws new masks every replay and the backward regenerates the mask instead of storing it.  They are not torch's Philox stream.
With p = 0 (every golden of tests/golden is generated that way) the result is exactly LayerNorm(x + sub).

``GVL_TRAIN_LAYERS=torch`` keeps the PyTorch formulation (A/B switch)."""
import ctypes
import itertools
import os

import torch
import torch.nn.functional as F

from . import _lib

_STEP = {}                       # device -> int64 counter tensor
_SITES = itertools.count(1)


def enabled():
    return os.environ.get("GVL_TRAIN_LAYERS", "") != "torch"


def step_counter(device):
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    t = _STEP.get(device)
    if t is None:
        t = _STEP[device] = torch.zeros(1, dtype=torch.int64, device=device)
    return t


class _ZeroArena:
    """zero-initialised scratch of ONE training step (row-maxima vectors that kernels fill with atomic max): one fill per step for
    all of them instead of one per vector.  `reset` opens a step (PDVC.forward, training) with a FRESH zero tensor as large as the
    last step asked for -- vectors an earlier step's autograd nodes still hold keep their storage --; without an open step, or past
    the capacity, `zeros` returns a torch.zeros of its own.  Graph-capturable (the allocation comes from the graph's pool)."""

    def __init__(self):
        self.buf, self.off, self.want, self.cap = None, 0, 0, 0
        self.snap = None                 # (step_snapshot below)

    def reset(self, device):
        self.cap = max(self.cap, self.want)
        self.buf = torch.zeros(self.cap, dtype=torch.float32, device=device) if self.cap else None
        self.off, self.want = 0, 0
        self.snap = None

    def zeros(self, n, device):
        need = (n + 63) // 64 * 64
        self.want += need
        if self.buf is None or self.buf.device != torch.device(device) or self.off + need > self.buf.numel():
            return torch.zeros(n, dtype=torch.float32, device=device)
        v = self.buf[self.off:self.off + n]
        self.off += need
        return v


_ARENA = {}


def _norm_device(device):
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def arena(device):
    device = _norm_device(device)
    a = _ARENA.get(device)
    if a is None:
        a = _ARENA[device] = _ZeroArena()
    return a


def arena_reset(device):
    device = _norm_device(device)
    arena(device).reset(device)


def step_zeros(n, device):
    """n fp32 zeros from the training step's arena (see _ZeroArena)"""
    device = _norm_device(device)
    return arena(device).zeros(n, device)


def step_snapshot(device):
    """the step counter's CURRENT value as a tensor of its own, for nodes whose backward must redraw the masks of their forward
    after the live counter has moved on: one copy per training step (taken at its first use after arena_reset / advance) shared
    by every such node, a copy per call when no step is open"""
    device = _norm_device(device)
    a = arena(device)
    if a.buf is None:
        return step_counter(device).clone()
    if a.snap is None:
        a.snap = step_counter(device).clone()
    return a.snap


def advance(device):
    """one training forward begins: the dropout masks of its residual chains change (graph-capturable)"""
    if torch.device(device).type != "cuda" or not enabled():
        return
    t = step_counter(device)
    arena(t.device).snap = None
    with torch.cuda.device(t.device):
        _lib.check(_lib.lib().gvl_advance_step(t.data_ptr(), torch.cuda.current_stream().cuda_stream), "advance_step")


def _site_seed(drop):
    site = drop.__dict__.get("_gvl_site")
    if site is None:
        site = drop.__dict__["_gvl_site"] = next(_SITES)
    return (torch.initial_seed() * 0x9E3779B1 + site * 0x85EBCA77) & 0xFFFFFFFF


def eligible(x, sub, norm, drop):
    return (enabled() and x.is_cuda and torch.is_grad_enabled() and not torch.is_autocast_enabled()
            and x.dtype == torch.float32 and sub.dtype == torch.float32 and x.dim() == 3 and x.shape == sub.shape
            and all(t_.stride(2) == 1 and t_.stride(0) % 4 == 0 and t_.stride(1) % 4 == 0 for t_ in (x, sub))
            and x.shape[-1] % 4 == 0 and x.shape[-1] <= 1024 and tuple(norm.normalized_shape) == (x.shape[-1],)
            and norm.elementwise_affine and norm.bias is not None and norm.weight.dtype == torch.float32
            and x.data_ptr() % 16 == 0 and sub.data_ptr() % 16 == 0 and x.numel() < 2 ** 32)


def _pos_strides(pos, B, Q, C):
    """(pointer, batch stride, query stride) of the positional addend of the NEXT attention query for k_rdln_fwd, or None when
    its layout is not addressable that way"""
    if pos is None or pos.dtype != torch.float32 or not pos.is_cuda or pos.dim() != 3 or tuple(pos.shape) != (B, Q, C) \
            or pos.stride(2) != 1 or pos.stride(0) % 4 or pos.stride(1) % 4 or pos.data_ptr() % 16:
        return None
    return pos.data_ptr(), pos.stride(0), pos.stride(1)


MAX_FAN = 6                 # gvl_rdln_backward_max_grads()


class ResidualDropoutLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sub, weight, bias, eps, p, seed, step, pos, fan=1):
        from . import layers as L
        B, Q, C = x.shape
        R = B * Q
        y = torch.empty(B, Q, C, device=x.device, dtype=torch.float32)
        z = torch.empty_like(y)
        stats = torch.empty(4, R, device=x.device, dtype=torch.float32)        # mean | rstd | max |y| | max |y + pos|
        # the step whose masks this forward drew, kept for the backward (the live counter may move before it runs: ADVICE r4)
        used = torch.empty(1, device=x.device, dtype=torch.int64) if step is not None else None
        ps = _pos_strides(pos, B, Q, C)
        with torch.cuda.device(x.device):
            rc = _lib.lib().gvl_residual_dropout_layer_norm_forward_f32(
                x.data_ptr(), x.stride(0), x.stride(1), sub.data_ptr(), sub.stride(0), sub.stride(1), Q, R, C,
                weight.data_ptr(), bias.data_ptr(),
                float(eps), float(p), int(seed), step.data_ptr() if step is not None else None, y.data_ptr(), z.data_ptr(),
                stats[0].data_ptr(), stats[1].data_ptr(), used.data_ptr() if used is not None else None,
                ps[0] if ps else None, ps[1] if ps else 0, ps[2] if ps else 0, stats[2].data_ptr(),
                stats[3].data_ptr() if ps else None, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "residual_dropout_layer_norm_forward")
        ctx.save_for_backward(z, stats, weight, used if used is not None else x.new_empty(0))
        ctx.p, ctx.seed, ctx.has_step = float(p), int(seed), step is not None
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)          # (the non-differentiable by-product would otherwise get a zeros() launch in backward)
        # fan > 1: the result once per consumer (aliases of one storage) -- each consumer's gradient arrives in its own slot and the
        # backward kernel sums them as it loads them, where autograd would add them with a launch per extra consumer
        ctx.fan = int(fan)
        return (y,) + tuple(y.view_as(y) for _ in range(ctx.fan - 1)) + (stats,)

    @staticmethod
    def backward(ctx, *grads):
        from . import layers as L
        z, stats, weight, step = ctx.saved_tensors
        B, Q, C = z.shape
        R = B * Q
        dys = [g.contiguous() for g in grads[:ctx.fan] if g is not None]
        if not dys:
            return (None,) * 10
        dy = dys[0]
        lib = _lib.lib()
        dz = torch.empty_like(z)
        dsub = torch.empty_like(z) if ctx.p > 0 else None
        part = torch.empty(lib.gvl_rdln_backward_blocks(R), 2 * C, device=z.device, dtype=torch.float32)
        dgb = torch.empty(2, C, device=z.device, dtype=torch.float32)
        am = torch.empty(R, device=z.device, dtype=torch.float32)
        with torch.cuda.device(z.device):
            rc = lib.gvl_residual_dropout_layer_norm_backwardn_f32(
                (ctypes.c_void_p * len(dys))(*[g.data_ptr() for g in dys]), len(dys), z.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), R, C, weight.data_ptr(), ctx.p,
                ctx.seed, step.data_ptr() if ctx.has_step else None, dz.data_ptr(),
                dsub.data_ptr() if dsub is not None else None, part.data_ptr(), dgb.data_ptr(), am.data_ptr(),
                torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "residual_dropout_layer_norm_backward")
        # (one bound serves both: |dsub| <= |dz| / (1 - p))
        L.tag_amax(dz, am)
        if dsub is not None:
            L.tag_amax(dsub, am)
        return dz, (dsub if dsub is not None else dz), dgb[0], dgb[1], None, None, None, None, None, None


def residual_dropout_norm(x, sub, drop, norm, pos=None, fan=1):
    """norm(x + drop(sub)) -- the fused kernels when they apply, PyTorch's formulation otherwise.  pos: the positional addend
    of the attention that consumes the result next; the result then also carries the row maxima of `result + pos`
    (add_pos below hands them to the sum), so the attention's projection needs no pass of its own over its query.
    fan = 2 .. MAX_FAN: a tuple of that many handles of the result, ONE PER CONSUMER (the next sublayer, the next residual, ...): on the
    fused path they are aliases whose gradients the backward kernel sums in its load path; otherwise the same tensor repeated"""
    p = drop.p if drop.training else 0.0
    if not eligible(x, sub, norm, drop) or p >= 1.0:
        y = norm(x + drop(sub))
        return y if fan == 1 else (y,) * fan
    from . import layers as L
    assert 1 <= fan <= MAX_FAN
    if fan > 1 and os.environ.get("GVL_RDLN_FAN", "1") == "0":               # (A/B switch: one handle, autograd adds the gradients)
        return (residual_dropout_norm(x, sub, drop, norm, pos),) * fan
    *ys, stats = ResidualDropoutLayerNorm.apply(x, sub, norm.weight, norm.bias, norm.eps, p, _site_seed(drop),
                                                step_counter(x.device) if p > 0 else None, pos, fan)
    for y in ys:
        L.tag_amax(y, stats[2])
        if _pos_strides(pos, *x.shape) is not None:
            y._gvl_amax_pos = (stats[3], y._version, pos.data_ptr(), pos._version)
    return ys[0] if fan == 1 else tuple(ys)


def add_pos(t, pos):
    """t + pos (with_pos_embed of pdvc/deformable_transformer.py:185,253) -- carrying the row maxima of the sum when t's producer
    left them (residual_dropout_norm(..., pos=pos))"""
    if pos is None:
        return t
    q = t + pos
    hit = getattr(t, "_gvl_amax_pos", None)
    if hit is not None and hit[1] == t._version and hit[2] == pos.data_ptr() and hit[3] == pos._version:
        from . import layers as L
        L.tag_amax(q, hit[0])
    return q


class ReluDropout(torch.autograd.Function):
    """dropout(relu(x)) in ONE kernel; backward from the OUTPUT alone (y > 0 exactly where the element was kept and positive):
    no mask tensor, and x -- the output of linear1, which no backward reads -- is free as soon as y exists.  Row forms of the
    kernels: the row maxima of y / of dx are left for the Linear products on either side."""

    @staticmethod
    def forward(ctx, x, p, seed, step):
        C = x.shape[-1]
        R = x.numel() // C
        y = torch.empty_like(x)
        am = torch.empty(R, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            rc = _lib.lib().gvl_relu_dropout_rows_forward_f32(x.data_ptr(), R, C, float(p), int(seed),
                                                              step.data_ptr() if step is not None else None, y.data_ptr(),
                                                              am.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "relu_dropout_forward")
        ctx.save_for_backward(y)
        ctx.p = float(p)
        ctx.mark_non_differentiable(am)
        ctx.set_materialize_grads(False)          # (the non-differentiable by-product would otherwise get a zeros() launch in backward)
        return y, am

    @staticmethod
    def backward(ctx, dy, _dam):
        from . import layers as L
        y, = ctx.saved_tensors
        dy = dy.contiguous()
        C = y.shape[-1]
        R = y.numel() // C
        dx = torch.empty_like(y)
        am = torch.empty(R, device=y.device, dtype=torch.float32)
        with torch.cuda.device(y.device):
            rc = _lib.lib().gvl_relu_dropout_rows_backward_f32(dy.data_ptr(), y.data_ptr(), R, C, ctx.p, dx.data_ptr(),
                                                               am.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "relu_dropout_backward")
        return L.tag_amax(dx, am), None, None, None


def relu_dropout(x, activation, drop):
    """drop(activation(x)) -- one kernel for ReLU on fp32 activations in training"""
    if not (enabled() and x.is_cuda and torch.is_grad_enabled() and not torch.is_autocast_enabled() and x.requires_grad
            and x.dtype == torch.float32 and x.is_contiguous() and x.dim() >= 2 and x.shape[-1] % 4 == 0 and x.numel() < 2 ** 32
            and x.data_ptr() % 16 == 0 and activation in (F.relu, torch.relu)
            and 0.0 <= (drop.p if drop.training else 0.0) < 1.0):
        return drop(activation(x))
    p = drop.p if drop.training else 0.0
    from . import layers as L
    y, am = ReluDropout.apply(x, p, _site_seed(drop), step_counter(x.device) if p > 0 else None)
    return L.tag_amax(y, am)
