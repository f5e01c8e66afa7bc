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


class _VocabNLLFunction(torch.autograd.Function):
    """weight[r] * log_softmax(x W^T + b)[r, target[r]] for every (caption row, token step) r, without the (R, V)
    log-prob tensor: product (fp16 matrix cores at fp32 accuracy when in the kernel's domain), one pass for log-sum-exp
    and the picked logit; backward the logits buffer is overwritten by its own gradient and feeds the three gradient
    products of the layer (include/gvl_msda.h: gvl_ce_rows_*_f32)."""

    @staticmethod
    def forward(ctx, x, weight, bias, target, row_weight):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        if split_gemm_enabled() and x2.shape[1] % 32 == 0 and weight.is_contiguous():
            logits = MSDA.gemm_f16x3(MSDA.split_rows(x2), MSDA.split_rows(weight.detach()), bias.detach())
        else:
            logits = torch.addmm(bias.detach(), x2, weight.detach().t())
        out, lse = MSDA.ce_rows_forward(logits, target, row_weight)
        ctx.save_for_backward(x2, weight, logits, lse, target, row_weight)
        ctx.x_shape = x.shape
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        x2, weight, logits, lse, target, row_weight = ctx.saved_tensors
        if getattr(ctx, "consumed", False):                 # the logits buffer now holds their gradient
            raise RuntimeError("vocab_nll: backward runs once (the logits are overwritten by their gradient); "
                               "GVL_VOCAB_NLL=torch keeps the log-prob formulation for retain_graph use")
        ctx.consumed = True
        g = MSDA.ce_rows_backward_(logits, target, row_weight, grad_out.contiguous(), lse)     # (R, V), in place
        gx = g.mm(weight).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        gw = g.t().mm(x2) if ctx.needs_input_grad[1] else None
        gb = MSDA.col_sum(g) if ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, None


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


def linear(x, weight, bias=None):
    if (bias is None or not x.is_cuda or x.dtype != torch.float32 or weight.dtype != torch.float32
            or bias.dtype != torch.float32 or torch.is_autocast_enabled()
            or not torch.is_grad_enabled() or not bias.requires_grad):
        return F.linear(x, weight, bias)
    return _LinearFunction.apply(x, weight, bias)


class Linear(nn.Linear):
    """Drop-in nn.Linear (same parameters, same state_dict keys)."""

    def forward(self, input):                                             # noqa: A002
        return linear(input, self.weight, self.bias)
