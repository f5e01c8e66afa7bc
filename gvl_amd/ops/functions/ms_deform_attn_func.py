"""Autograd boundary of the op -- mirrors pdvc/ops/functions/ms_deform_attn_func.py:23-41.

``MSDeformAttnFunction.apply(value, value_spatial_shapes, value_level_start_index, sampling_locations,
attention_weights, im2col_step)`` keeps the reference's signature, saved tensors, once_differentiable backward and
6-tuple of gradients; it computes the reference CUDA op's zero-padding semantics on the HIP kernels.
``MSDeformAttnPadFunction`` is the same op with an explicit ``pad_mode`` ("border" reproduces the arithmetic of the
reference's CPU fallback ms_deform_attn_core_pytorch, func.py:44-71, on the GPU).  There is deliberately no
PyTorch fallback here: without libgvl_msda.so or a ROCm device the call raises.
"""
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import torch

from ... import MultiScaleDeformableAttention as MSDA


def _arith(value, *tensors):
    """bf16 autocast: value may arrive in bf16 (a Linear output) while locations / weights are kept -- or brought back
    to -- fp32 (bf16 is a storage type for the op, see include/gvl_msda.h "Element types")."""
    if value.dtype != torch.bfloat16:
        return tensors
    return tuple(t_ if t_.dtype == torch.float32 else t_.float() for t_ in tensors)


def _like(grad, dtype):
    return grad if grad is None or grad.dtype == dtype else grad.to(dtype)


class MSDeformAttnPadFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step, pad_mode):
        ctx.im2col_step = im2col_step
        ctx.pad_mode = pad_mode
        ctx.host = MSDA.host_shapes(value_spatial_shapes, value_level_start_index)
        ctx.in_dtypes = (sampling_locations.dtype, attention_weights.dtype)
        sampling_locations, attention_weights = _arith(value, sampling_locations, attention_weights)
        output = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                             sampling_locations, attention_weights, im2col_step, pad_mode)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, attn = ctx.saved_tensors
        if getattr(shapes, "_gvl_host", None) is None:
            shapes._gvl_host = ctx.host
        gv, gl, ga = MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, attn,
                                                  _like(grad_output, value.dtype).contiguous(), ctx.im2col_step,
                                                  ctx.pad_mode)
        return gv, None, None, _like(gl, ctx.in_dtypes[0]), _like(ga, ctx.in_dtypes[1]), None, None


class MSDeformAttnFunction(Function):
    """Exact reference signature (func.py:23-41); zero padding == the reference CUDA kernels."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        ctx.host = MSDA.host_shapes(value_spatial_shapes, value_level_start_index)
        ctx.in_dtypes = (sampling_locations.dtype, attention_weights.dtype)
        sampling_locations, attention_weights = _arith(value, sampling_locations, attention_weights)
        output = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                             sampling_locations, attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, attn = ctx.saved_tensors
        if getattr(shapes, "_gvl_host", None) is None:
            shapes._gvl_host = ctx.host
        grad_value, grad_sampling_loc, grad_attn_weight = MSDA.ms_deform_attn_backward(
            value, shapes, lsi, loc, attn, _like(grad_output, value.dtype).contiguous(), ctx.im2col_step)
        return (grad_value, None, None, _like(grad_sampling_loc, ctx.in_dtypes[0]),
                _like(grad_attn_weight, ctx.in_dtypes[1]), None)


class MSDASampleFunction(Function):
    """Differentiable unweighted sampler: ms_deform_attn_core_pytorch(..., return_value=True) (func.py:44-68) with
    the gradients F.grid_sample would give (w.r.t. value and sampling locations)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, pad_mode):
        ctx.pad_mode = pad_mode
        ctx.host = MSDA.host_shapes(value_spatial_shapes, value_level_start_index)
        ctx.in_dtypes = (value.dtype, sampling_locations.dtype)
        if value.dtype == torch.bfloat16:            # D = 512 sampler: widened (fp32 arithmetic), rounded once
            value, sampling_locations = value.float(), sampling_locations.float()
        out = MSDA.ms_deform_attn_sample(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                         pad_mode)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations)
        return _like(out, ctx.in_dtypes[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_sample):
        value, shapes, lsi, loc = ctx.saved_tensors
        gv, gl = MSDA.ms_deform_attn_sample_backward(value, shapes, lsi, loc,
                                                     _like(grad_sample, value.dtype).contiguous(), ctx.pad_mode)
        return _like(gv, ctx.in_dtypes[0]), None, None, _like(gl, ctx.in_dtypes[1]), None


class MSDeformAttnFusedFunction(Function):
    """MSDeformAttn.forward between its projection GEMM and output_proj (ms_deform_attn.py:99-124) as ONE op:
    (value, proj = [raw offsets | attention logits], reference points) -> sampled output.  The softmax, the
    location arithmetic and their backward live inside the HIP kernels; loc / attn never exist in HBM."""

    @staticmethod
    def forward(ctx, value, proj, reference_points, spatial_shapes, level_start_index, n_levels, n_points, pad_mode,
                want_amax=False):
        ctx.cfg = (n_levels, n_points, pad_mode)
        ctx.host = MSDA.host_shapes(spatial_shapes, level_start_index)
        ctx.in_dtypes = (proj.dtype, reference_points.dtype)
        proj = _like(proj, value.dtype)                              # bf16 value <-> bf16 projection rows
        reference_points = _like(reference_points, torch.float32)    # positions are never rounded to bf16
        # fp32, want_amax (the caller is a training forward; grad mode is OFF in here whatever the caller's, so it cannot be asked):
        # the kernel also leaves max |out row| (one atomic max per row and head) -- the row scale output_proj's split needs
        amax = None
        if value.dtype == torch.float32 and want_amax:
            from ... import train_layers as _tl
            amax = _tl.step_zeros(proj.shape[0] * proj.shape[1], value.device)
        out = MSDA.msda1d_fused_forward(value, spatial_shapes, level_start_index, proj, reference_points, n_levels,
                                        n_points, pad_mode, amax_out=amax)
        ctx.save_for_backward(value, proj, reference_points, spatial_shapes, level_start_index)
        if amax is not None:
            ctx.mark_non_differentiable(amax)
            ctx.set_materialize_grads(False)          # (the non-differentiable by-product would otherwise get a zeros() launch in backward)
            return out, amax
        return out, None

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output, _damax=None):
        value, proj, ref, shapes, lsi = ctx.saved_tensors
        if getattr(shapes, "_gvl_host", None) is None:
            shapes._gvl_host = ctx.host
        n_levels, n_points, pad_mode = ctx.cfg
        gv, gp, gr = MSDA.msda1d_fused_backward(value, shapes, lsi, proj, ref,
                                                _like(grad_output, value.dtype).contiguous(), n_levels, n_points,
                                                pad_mode, need_ref_grad=ctx.needs_input_grad[2])
        return gv, _like(gp, ctx.in_dtypes[0]), _like(gr, ctx.in_dtypes[1]), None, None, None, None, None, None


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights, return_value=False):
    """Same name, arguments and results as the reference's pure-PyTorch core (func.py:44-71: per-level bilinear
    ``grid_sample`` with ``padding_mode='border'``, optionally returning the unweighted samples (B*M, D, Lq, L, P)) --
    but computed by the HIP kernels on the GPU, differentiable w.r.t. value / locations / weights.  It is NOT a CPU
    fallback: CPU tensors raise, like every other entry point of gvl_amd."""
    import torch
    if not value.is_cuda:
        raise RuntimeError("gvl_amd.ms_deform_attn_core_pytorch runs on a ROCm device only (no CPU fallback)")
    shapes = value_spatial_shapes.to(device=value.device, dtype=torch.int64).contiguous()
    if shapes.dim() == 1:
        shapes = torch.stack([torch.ones_like(shapes), shapes], -1).contiguous()
    sizes = shapes[:, 0] * shapes[:, 1]
    lsi = torch.cat((sizes.new_zeros(1), sizes.cumsum(0)[:-1])).contiguous()
    value = value.contiguous()
    loc = sampling_locations.contiguous()
    if return_value:
        return MSDASampleFunction.apply(value, shapes, lsi, loc, "border")
    return MSDeformAttnPadFunction.apply(value, shapes, lsi, loc, attention_weights.contiguous(), 1 << 30, "border")
