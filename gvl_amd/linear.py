"""nn.Linear whose bias gradient is one column-sum kernel (include/gvl_msda.h: gvl_col_sum_f32).

autograd differentiates F.linear as two GEMMs plus `grad_output.sum(0)`; PyTorch's generic reduction over the slow
axis takes 10-20 us on the (B*Q | B*S, 256...2048) gradients of the path, ~40 times per train step.  `linear()` keeps
the two GEMMs exactly as autograd issues them (same operand order, so the same tuned library kernels) and replaces
only that reduction.  Anything it is not built for -- no bias, CPU, non-fp32, autocast -- goes through
F.linear unchanged."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import MultiScaleDeformableAttention as MSDA


class _LinearFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight, bias)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        g2 = grad_out.reshape(-1, grad_out.shape[-1])
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = g2.mm(weight).view(x.shape)                              # AddmmBackward: grad @ mat2^T
        if ctx.needs_input_grad[1]:
            # AddmmBackward sees mat2 = weight.t() (column-major) and takes the branch grad^T @ mat1: the gradient
            # arrives contiguous in the weight's own layout and AccumulateGrad keeps it without a copy
            gw = g2.t().mm(x.reshape(-1, x.shape[-1]))
        if ctx.needs_input_grad[2]:
            gb = MSDA.col_sum(g2) if MSDA.col_sum_eligible(g2) else g2.sum(0)
        return gx, gw, gb


class _ProjFunction(_LinearFunction):
    """the same linear map with the FORWARD product on the hand-written fp32 MFMA kernel (include/gvl_msda.h:
    gvl_proj_f32); the two gradient GEMMs stay the library's, the bias gradient is gvl_col_sum_f32"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return MSDA.proj_linear(x, weight, bias)


def projection(x, weight, bias):
    """MSDeformAttn's offset / attention-logit projection (ms_deform_attn.py:99-100 against the concatenated weight) in
    TRAINING (inference runs it as a column segment of gvl_linear_f16x3_f32, gvl_amd/layers.py).  Default: the tuned library
    GEMM through linear() -- the hand-written exact-fp32 MFMA kernel gvl_proj_f32 measures 17.2 us against the library's
    16.7 us at R = 4800 in the same trace (profiles/r03_train_kernel_stats.txt, profiles/r02_proj_gemm.txt), so it is no
    longer the default; GVL_PROJ=own selects it (contiguous fp32, K in {256, 512, 1024}, N % 64 == 0,
    no autocast) and tests/test_gpu_criterion.py keeps it pinned."""
    if (os.environ.get("GVL_PROJ", "") != "own" or torch.is_autocast_enabled()
            or not MSDA.proj_eligible(x, weight, bias)):
        return linear(x, weight, bias)
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        return _ProjFunction.apply(x, weight, bias)
    return MSDA.proj_linear(x, weight, bias)


def split_gemm_enabled():
    """the captioner's fp32 token-loop products run on the fp16 matrix cores at fp32 accuracy (include/gvl_msda.h:
    gvl_gemm_f16x3_f32) unless GVL_GEMM=f32 asks for the fp32 library GEMMs (A/B runs)"""
    return os.environ.get("GVL_GEMM", "") != "f32"


class _SplitLinearFunction(_LinearFunction):
    """the same linear map with the FORWARD product on the fp16 matrix cores at fp32 accuracy (include/gvl_msda.h:
    gvl_split_rows_f16 + gvl_gemm_f16x3_f32); the gradients are _LinearFunction's"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        out = MSDA.gemm_f16x3(MSDA.split_rows(x2), MSDA.split_rows(weight.detach()), bias.detach())
        return out.view(*x.shape[:-1], weight.shape[0])


def split_linear(x, weight, bias):
    """nn.Linear for a LARGE fp32 product in training (the vocabulary layer over all teacher-forced steps, 2208 x 512 x
    8518 at cfg A: 310 us as an fp32 library GEMM, ~100 us here including the split of both operands); anything outside
    the kernel's domain goes through linear()"""
    if (bias is None or torch.is_autocast_enabled() or not split_gemm_enabled() or not x.is_cuda
            or x.dtype != torch.float32 or weight.dtype != torch.float32 or bias.dtype != torch.float32
            or x.shape[-1] % 32 or not weight.is_contiguous() or not bias.is_contiguous()
            or x.numel() // x.shape[-1] < 1024):
        return linear(x, weight, bias)
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or bias.requires_grad):
        return _SplitLinearFunction.apply(x, weight, bias)
    return _SplitLinearFunction.forward(_NoCtx(), x, weight, bias)


@MSDA.keeps_products
class _VocabNLLFunction(torch.autograd.Function):
    """weight[r] * log_softmax(x W^T + b)[r, target[r]] for every (caption row, token step) r, without the (R, V)
    log-prob tensor: product (fp16 matrix cores at fp32 accuracy when in the kernel's domain), one pass for log-sum-exp
    and the picked logit; backward the logits buffer is overwritten by its own gradient and feeds the three gradient
    products of the layer (include/gvl_msda.h: gvl_ce_rows_*_f32)."""

    @staticmethod
    def forward(ctx, x, weight, bias, target, row_weight):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        R, V = x2.shape[0], weight.shape[0]
        # rows of the logits padded to a whole K stage of 32 columns: the hand-written gradient products read whole stages of a row
        # (the padding is zeroed by gvl_ce_rows_backward_f32)
        ld = (V + 31) // 32 * 32
        logits = torch.empty(R, ld, device=x.device, dtype=torch.float32)[:, :V]
        planes = _vocab_planes(weight, bias)
        if split_gemm_enabled() and x2.shape[1] % 32 == 0 and weight.is_contiguous():
            MSDA.gemm_f16x3(MSDA.split_rows(x2), planes[0] if planes is not None else MSDA.split_rows(weight.detach()),
                            bias.detach(), out=logits)
        else:
            torch.addmm(bias.detach(), x2, weight.detach().t(), out=logits)
        out, lse = MSDA.ce_rows_forward(logits, target, row_weight)
        ctx.save_for_backward(x2, weight, logits, lse, target, row_weight)
        ctx.x_shape, ctx.planes_t = x.shape, planes[1] if planes is not None else None
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        from . import layers as L
        from .train_planes import Operand
        x2, weight, logits, lse, target, row_weight = ctx.saved_tensors
        if getattr(ctx, "consumed", False):                 # the logits buffer now holds their gradient
            raise RuntimeError("vocab_nll: backward runs once (the logits are overwritten by their gradient); "
                               "GVL_VOCAB_NLL=torch keeps the log-prob formulation for retain_graph use")
        ctx.consumed = True
        own = _TRAIN_LINEAR and MSDA.wgrad_eligible(logits, x2)
        am_g = torch.empty(logits.shape[0], device=logits.device, dtype=torch.float32) if own else None
        g = MSDA.ce_rows_backward_(logits, target, row_weight, grad_out.contiguous(), lse, amax=am_g)     # (R, V), in place
        gx = None
        if ctx.needs_input_grad[0]:
            K = x2.shape[1]
            if own and ctx.planes_t is not None and K % 128 == 0:
                # dx = g W on the planes of W^T, the contraction over the 8518 vocabulary entries cut into K-stage ranges
                # (gvl_linear_f16x3_splitk_f32): 72 output tiles alone would leave three quarters of the chip idle
                g_pad = g.as_strided((g.shape[0], g.stride(0)), (g.stride(0), 1))        # the rows with their zeroed padding
                gx = L.linear_splitk(g_pad, am_g, Operand(ctx.planes_t, K, g.stride(0), None)).view(ctx.x_shape)
            else:
                gx = g.mm(weight).view(ctx.x_shape)
        if own and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            # dW = g^T x and db = sum_r g in ONE pass over g on the fp16 matrix cores (gvl_wgrad_f16x3_f32): 135 us against the
            # library's 241 us + the column sum's 17 us at (2208, 8518, 512)
            # (the rows of padded positions are zero, and say so in am_g: the product runs over the list of the other rows)
            gw, gb = MSDA.wgrad(g, x2, am_g, L.row_absmax(x2)[0], skip_zero_rows=os.environ.get("GVL_WGRAD_SKIP", "1") != "0")
            return gx, gw, gb, None, None
        gw = g.t().mm(x2) if ctx.needs_input_grad[1] else None
        gb = MSDA.col_sum(g) if ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, None


def _vocab_planes(weight, bias):
    """(planes of W, planes of W^T) of the vocabulary layer from the active TrainPlanes (refreshed for this forward), or None"""
    tp = _ACTIVE_PLANES
    if tp is None or not _TRAIN_LINEAR:
        return None
    hit = tp.lookup((weight,))
    if hit is None or not tp.is_fresh():
        return None
    return hit[0], hit[1]


def vocab_nll_eligible(x, weight, bias):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and bias is not None
            and bias.dtype == torch.float32 and not torch.is_autocast_enabled() and bias.is_contiguous()
            and os.environ.get("GVL_VOCAB_NLL", "") != "torch")


def vocab_nll(x, weight, bias, target, row_weight):
    """(..., H) hidden states -> (R,) row_weight * log_softmax(linear(x))[target] (R = product of the leading dimensions);
    callers check vocab_nll_eligible first"""
    return _VocabNLLFunction.apply(x, weight, bias, target.reshape(-1).contiguous(),
                                   row_weight.reshape(-1).to(torch.float32).contiguous())


class _NoCtx:
    def save_for_backward(self, *a):
        pass


# ---- the TRAINING form on the hand-written kernels -------------------------------------------------------------------------
# forward  y = x [W_0; W_1; ..]^T + [b_0; b_1; ..]   gvl_linear_f16x3_f32 on the planes of the stacked weight
# backward dx = dy W                                  the same kernel on the planes of the TRANSPOSED stacked weight
#          dW = dy^T x,  db = sum_r dy                gvl_wgrad_f16x3_f32 (one pass over dy for both)
# The planes come from the model's TrainPlanes (gvl_amd/train_planes.py: all weights of the step in two launches per forward);
# a weight that is not registered there is split on demand, per parameter version.  The row maxima the activation splits need
# ride on the tensors (gvl_amd.layers.tag_amax) when the producer is one of the path's own kernels; otherwise one
# gvl_row_absmax_f32 launch computes them.
_TRAIN_LINEAR = os.environ.get("GVL_TRAIN_LINEAR", "") != "torch"
_ACTIVE_PLANES = None
MIN_TRAIN_ROWS = 512


def train_linear_enabled(on=None):
    """query / set the switch (GVL_TRAIN_LINEAR=torch at import: the library GEMMs of round 4, for A/B runs)"""
    global _TRAIN_LINEAR
    if on is not None:
        _TRAIN_LINEAR = bool(on)
    return _TRAIN_LINEAR


def set_active_planes(planes):
    """the TrainPlanes the current training forward refreshed (gvl_amd.pdvc.PDVC.forward), or None"""
    global _ACTIVE_PLANES
    prev, _ACTIVE_PLANES = _ACTIVE_PLANES, planes
    return prev


class _WgradQueue:
    """Weight gradients of the layers' Linears, owed during one backward pass and taken TOGETHER (gvl_wgrad_group_f16x3_f32: up to
    ten gradients per launch -- 52 against 120 us for an encoder layer's five, 130 against 216 us for a decoder layer's eight,
    tools/wgrad_group_probe.py).  Nobody reads dW before the optimizer, so a Linear's backward hands autograd the (still unwritten)
    gradient tensors and the products run when the group is full or the pass ends.  What makes that safe:
      * only Linears that own their weight alone are queued (`defer`: the encoder / decoder layers'); and should a parameter
        get a second gradient in the same pass after all, autograd would ADD the two the moment the second is returned -- the
        queue is flushed and that product taken at once;
      * a parameter whose .grad already exists (gradient accumulation, the data-parallel flat buffer) is never queued:
        AccumulateGrad reads the new gradient immediately;
      * the queue keeps detached aliases: autograd finds the returned tensor unshared and adopts it without a copy."""

    def __init__(self):
        self.items, self.seen = [], set()

    def push(self, dy, x, am_dy, am_x, want_bias, keys):
        gw = torch.empty(dy.shape[1], x.shape[1], device=dy.device, dtype=torch.float32)
        gb = torch.empty(dy.shape[1], device=dy.device, dtype=torch.float32) if want_bias else None
        if any(k in self.seen for k in keys):
            self.flush()
            MSDA.wgrad(dy, x, am_dy, am_x, grad_w=gw, grad_b=gb, want_bias=want_bias)
            return gw, gb
        if len(self.items) >= MSDA.wgrad_group_max():                 # (another node may have filled the group exactly: _InProj)
            self.flush()
        self.seen.update(keys)
        self.items.append((dy, x, am_dy, am_x, gw.detach(), gb.detach() if gb is not None else None))
        if len(self.items) >= MSDA.wgrad_group_max():
            self.flush()
        return gw, gb

    def flush(self):
        items, self.items = self.items, []
        if len(items) == 1:
            dy, x, am_dy, am_x, gw, gb = items[0]
            MSDA.wgrad(dy, x, am_dy, am_x, grad_w=gw, grad_b=gb, want_bias=gb is not None)
        elif items:
            MSDA.wgrad_group(items)


_WGRAD_QUEUE = None
_DEFER_WGRAD = os.environ.get("GVL_WGRAD_GROUP", "") != "0"
# products over this many rows or more are taken one by one (long videos: the encoder's 15 360 rows at T = 512 -- a launch of its
# own already fills the chip, and the group's two co-resident workgroups per CU lose to it: tools/switch_sweep.sh)
_GROUP_MAX_ROWS = int(os.environ.get("GVL_WGRAD_GROUP_MAX_ROWS", "8192"))


class deferred_wgrads:
    """``with deferred_wgrads(): loss.backward()`` -- the layers' weight gradients of this backward pass in grouped launches
    (see _WgradQueue); flushed on exit.  GVL_WGRAD_GROUP=0: one launch pair per Linear, as in round 5 (A/B switch)."""

    def __enter__(self):
        global _WGRAD_QUEUE
        self.prev = _WGRAD_QUEUE
        _WGRAD_QUEUE = _WgradQueue() if _DEFER_WGRAD else None
        return self

    def __exit__(self, exc_type, exc, tb):
        global _WGRAD_QUEUE
        q, _WGRAD_QUEUE = _WGRAD_QUEUE, self.prev
        if q is not None and exc_type is None:
            q.flush()
        return False


def queued_wgrad(dy, x, am_dy, am_x, want_bias, params, defer):
    """(grad_w, grad_b) of one Linear: through the active queue when `defer` and every parameter's .grad is still None, else now"""
    q = _WGRAD_QUEUE
    if defer and q is not None and dy.shape[0] < _GROUP_MAX_ROWS and all(p_ is None or p_.grad is None for p_ in params):
        return q.push(dy, x, am_dy, am_x, want_bias, [id(p_) for p_ in params if p_ is not None])
    return MSDA.wgrad(dy, x, am_dy, am_x, want_bias=want_bias)


def _operands(weights, biases):
    """(forward operand, transposed operand) of the stacked weight"""
    from .train_planes import Operand
    tp = _ACTIVE_PLANES
    n_total, K = sum(w.shape[0] for w in weights), weights[0].shape[1]
    if tp is not None:
        hit = tp.lookup(weights)
        if hit is not None and tp.is_fresh():
            return Operand(hit[0], n_total, K, hit[2]), Operand(hit[1], K, n_total, None)
    # not registered with the active planes (a module used on its own): a private TrainPlanes for this operand, refreshed when one
    # of its parameters changes
    owner = weights[0]
    cached = owner.__dict__.get("_gvl_train_planes")
    if cached is None or cached.key_of(weights) not in cached.by_key or cached.device != owner.device:
        from .train_planes import TrainPlanes
        cached = owner.__dict__["_gvl_train_planes"] = TrainPlanes(owner.device)
        cached.register([w for w in weights], [b for b in biases])
    # (under stream capture ALWAYS: a forward captured right after an eager one at the same parameter versions would otherwise
    #  record no refresh, and every replay would multiply by the planes of the capture-time weights)
    if not cached.is_fresh() or torch.cuda.is_current_stream_capturing():
        with torch.no_grad():
            cached.refresh()
    hit = cached.lookup(weights)
    return Operand(hit[0], n_total, K, hit[2]), Operand(hit[1], K, n_total, None)


def _row_amax(t2, src=None):
    """row maxima of the (R, C) matrix t2: the tag its producer left (on t2 itself or on the tensor `src` it is a view of), or one
    gvl_row_absmax_f32 launch"""
    from . import layers as L
    for cand in (t2, src):
        if cand is not None:
            am = L.amax_of(cand, t2.shape[0])
            if am is not None:
                return am
    return L.row_absmax(t2)[0]


@MSDA.keeps_products
class _TrainLinearFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nblk, defer, *params):
        from . import layers as L
        weights, biases = params[:nblk], params[nblk:]
        ctx.defer = bool(defer)
        K = weights[0].shape[1]
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        am = _row_amax(x2, x)
        op, op_t = _operands(weights, biases)
        # (allocated in the shape it is returned in: a view created in here could not be edited in place by the caller --
        #  gvl_amd.layers.mask_rows zeroes the padded rows of a value projection that way)
        out = torch.empty(*x.shape[:-1], op.N, device=x.device, dtype=torch.float32)
        L.linear(x2, op, [L.seg(0, out.view(x2.shape[0], op.N), am)])
        ctx.save_for_backward(x2, am, *weights)
        ctx.op_t, ctx.nblk, ctx.x_shape = op_t, nblk, x.shape
        ctx.has_bias = [b is not None for b in biases]
        ctx.bias_params = tuple(biases)                       # (identity only: which parameters this node's gradients go to)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        from . import layers as L
        x2, am_x = ctx.saved_tensors[:2]
        weights = ctx.saved_tensors[2:]
        nblk, op_t = ctx.nblk, ctx.op_t
        n_total = op_t.K
        g2 = grad_out.reshape(-1, n_total)
        if g2.stride(1) != 1 or g2.stride(0) % 4 or g2.data_ptr() % 16:
            g2 = g2.contiguous()
        am_g = _row_amax(g2, grad_out)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(x2.shape[0], op_t.N, device=g2.device, dtype=torch.float32)
            L.linear(g2, op_t, [L.seg(0, gx, am_g)])
            gx = gx.view(ctx.x_shape)
        need_w = any(ctx.needs_input_grad[3:3 + nblk])
        need_b = any(n and h for n, h in zip(ctx.needs_input_grad[3 + nblk:], ctx.has_bias))
        gws, gbs = [None] * nblk, [None] * nblk
        if need_w or need_b:
            gw, gb = queued_wgrad(g2, x2, am_g, am_x, need_b, tuple(weights) + tuple(ctx.bias_params), ctx.defer)
            off = 0
            for i, w in enumerate(weights):
                n = w.shape[0]
                if ctx.needs_input_grad[3 + i]:
                    gws[i] = gw[off:off + n]
                if ctx.has_bias[i] and ctx.needs_input_grad[3 + nblk + i]:
                    gbs[i] = gb[off:off + n]
                off += n
        return (gx, None, None, *gws, *gbs)


@MSDA.keeps_products
class _MirrorLinearFunction(torch.autograd.Function):
    """x W_block^T for a COLUMN BLOCK of a parameter (w_view: strided) whose contiguous mirror `buf` is an operand of the active
    TrainPlanes (train_planes.register_mirror): the products run on the mirror's planes, the weight gradient goes to the view"""

    @staticmethod
    def forward(ctx, x, w_view, buf):
        from . import layers as L
        K = buf.shape[1]
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        am = _row_amax(x2, x)
        op, op_t = _operands((buf,), (None,))
        out = torch.empty(*x.shape[:-1], op.N, device=x.device, dtype=torch.float32)
        L.linear(x2, op, [L.seg(0, out.view(x2.shape[0], op.N), am)])
        ctx.save_for_backward(x2, am)
        ctx.op_t, ctx.x_shape = op_t, x.shape
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        from . import layers as L
        x2, am_x = ctx.saved_tensors
        op_t = ctx.op_t
        g2 = grad_out.reshape(-1, op_t.K)
        if g2.stride(1) != 1 or g2.stride(0) % 4 or g2.data_ptr() % 16:
            g2 = g2.contiguous()
        am_g = _row_amax(g2, grad_out)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(x2.shape[0], op_t.N, device=g2.device, dtype=torch.float32)
            L.linear(g2, op_t, [L.seg(0, gx, am_g)])
            gx = gx.view(ctx.x_shape)
        gw = MSDA.wgrad(g2, x2, am_g, am_x, want_bias=False)[0] if ctx.needs_input_grad[1] else None
        return gx, gw, None


def mirror_linear_eligible(x, w_view, buf):
    tp = _ACTIVE_PLANES
    return (buf is not None and tp is not None and tuple(buf.shape) == tuple(w_view.shape) and tp.lookup((buf,)) is not None
            and tp.is_fresh() and train_linear_eligible(x, (buf,), (None,)) or False) and (x.requires_grad or w_view.requires_grad)


def mirror_linear(x, w_view, buf):
    return _MirrorLinearFunction.apply(x, w_view, buf)


def train_linear_eligible(x, weights, biases):
    if not (_TRAIN_LINEAR and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
            and not torch.is_autocast_enabled() and x.dim() >= 2):
        return False
    K = x.shape[-1]
    if K % 64 or x.numel() // K < MIN_TRAIN_ROWS or x.numel() >= 2 ** 31:
        return False
    n_total = 0
    for w, b in zip(weights, biases):
        if not (w.dtype == torch.float32 and w.dim() == 2 and w.shape[1] == K and w.shape[0] % 32 == 0 and w.is_contiguous()
                and w.is_cuda and (b is None or (b.dtype == torch.float32 and b.is_contiguous() and b.numel() == w.shape[0]))):
            return False
        n_total += w.shape[0]
    if n_total % 64:
        return False
    return x.requires_grad or any(w.requires_grad for w in weights) or any(b is not None and b.requires_grad for b in biases)


def train_linear(x, weights, biases, defer=False):
    """x (..., K) -> x [W_0; W_1; ..]^T + [b_0; ..]  (..., sum N_i) through the hand-written kernels, differentiable in x, every
    W_i and every b_i; callers check train_linear_eligible first.  defer: the weight / bias gradients may join the backward
    pass's grouped launches (_WgradQueue: only for parameters no other node of the step uses)."""
    return _TrainLinearFunction.apply(x, len(weights), defer, *weights, *biases)


def linear(x, weight, bias=None, defer=False):
    if train_linear_eligible(x, (weight,), (bias,)):
        return train_linear(x, (weight,), (bias,), defer)
    if (bias is None or not x.is_cuda or x.dtype != torch.float32 or weight.dtype != torch.float32
            or bias.dtype != torch.float32 or torch.is_autocast_enabled()
            or not torch.is_grad_enabled() or not bias.requires_grad):
        return F.linear(x, weight, bias)
    return _LinearFunction.apply(x, weight, bias)


class Linear(nn.Linear):
    """Drop-in nn.Linear (same parameters, same state_dict keys)."""

    defer_wgrad = False      # set by the owner of a Linear that nothing else in the step uses (DeformableTransformer's layers)

    def forward(self, input):                                             # noqa: A002
        return linear(input, self.weight, self.bias, self.defer_wgrad)
