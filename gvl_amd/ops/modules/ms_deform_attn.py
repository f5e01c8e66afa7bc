"""MSDeformAttn -- host-side mirror of pdvc/ops/modules/ms_deform_attn.py:30-126 on the HIP op.

Same constructor, parameter names / shapes (``sampling_offsets, attention_weights, value_proj, output_proj``) and
initialisation (ms_deform_attn.py:62-77), same forward signature and errors, so reference checkpoints load with
``strict=True``.  ``pad_mode`` selects the sampling semantics: "zeros" (default) is what the reference computes on
a GPU through MSDeformAttnFunction; "border" is what its CPU fallback computes.
"""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn

from .. import functions as _fn
from ... import MultiScaleDeformableAttention as MSDA
from ...linear import Linear, linear, projection, train_linear, train_linear_eligible


def _mask_rows(value, padding_mask):
    """value with its padded rows zeroed (:100) -- gvl_amd.layers.mask_rows (imported late: layers imports this package)"""
    from ... import layers
    return layers.mask_rows(value, padding_mask)


def _power_of_two(n):
    if not isinstance(n, int) or n < 0:
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return n != 0 and (n & (n - 1)) == 0


def offset_bias_init(n_heads, n_levels, n_points, centre=False):
    """ms_deform_attn.py:64-69: per head h the x-component of the unit-box direction at angle 2*pi*h/M, scaled by
    (p + 1) for point p, identical on every level; the captioner variant subtracts the mean over points
    (ms_deform_attn_for_caption.py:72)."""
    theta = torch.arange(n_heads, dtype=torch.float32) * (2.0 * math.pi / n_heads)
    cs = torch.stack([theta.cos(), theta.sin()], -1)
    gx = (cs / cs.abs().max(-1, keepdim=True)[0])[:, 0]                      # (M,)
    scale = torch.arange(1, n_points + 1, dtype=torch.float32)               # (P,)
    bias = gx[:, None, None] * scale[None, None, :] * torch.ones(1, n_levels, 1)
    if centre:
        bias = bias - bias.mean(2, keepdim=True)
    return bias.reshape(-1)


def temporal_shapes_2d(input_spatial_shapes, input_level_start_index):
    """(L,) lengths -> the (L,2) [(1,T_l)] tensor the op consumes (ms_deform_attn.py:117), carrying a host copy so
    the C ABI can pick the temporal kernels without a device->host read."""
    cached = getattr(input_spatial_shapes, "_gvl_shapes2d", None)
    if cached is not None:
        return cached
    shapes2d = torch.stack([torch.ones_like(input_spatial_shapes), input_spatial_shapes], -1).contiguous()
    host = getattr(input_spatial_shapes, "_gvl_host_lengths", None)
    if host is not None:
        lengths, starts = host
        MSDA.attach_host_shapes(shapes2d, input_level_start_index, [(1, int(t_)) for t_ in lengths], starts)
        input_spatial_shapes._gvl_shapes2d = shapes2d          # level tensors are cached constants
    return shapes2d


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, im2col_step=64, pad_mode="zeros"):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError('d_model must be divisible by n_heads, but got {} and {}'.format(d_model, n_heads))
        if not _power_of_two(d_model // n_heads):
            warnings.warn("You'd better set d_model in MSDeformAttn to make the dimension of each attention head a "
                          "power of 2 which is more efficient in our CUDA implementation.")
        self.im2col_step = im2col_step
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.pad_mode = pad_mode
        self.fused = True              # use the fused projection-epilogue + sampling kernels when the shape allows
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = Linear(d_model, d_model)
        self.output_proj = Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        with torch.no_grad():
            self.sampling_offsets.weight.zero_()
            self.sampling_offsets.bias.copy_(offset_bias_init(self.n_heads, self.n_levels, self.n_points))
            self.attention_weights.weight.zero_()
            self.attention_weights.bias.zero_()
            nn.init.xavier_uniform_(self.value_proj.weight)
            self.value_proj.bias.zero_()
            nn.init.xavier_uniform_(self.output_proj.weight)
            self.output_proj.bias.zero_()

    def _project(self, query, input_flatten, input_padding_mask):
        N, Len_q, _ = query.shape
        _, Len_in, _ = input_flatten.shape
        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = _mask_rows(value, input_padding_mask)
        value = value.view(N, Len_in, self.n_heads, self.d_model // self.n_heads)
        off = self.sampling_offsets(query).view(N, Len_q, self.n_heads, self.n_levels, self.n_points)
        aw = self.attention_weights(query).view(N, Len_q, self.n_heads, self.n_levels * self.n_points)
        aw = F.softmax(aw, -1).view(N, Len_q, self.n_heads, self.n_levels, self.n_points)
        return value, off, aw

    def _locations(self, reference_points, off, input_spatial_shapes):
        if reference_points.shape[-1] == 1:
            x = reference_points[:, :, None, :, None, 0] + off / input_spatial_shapes[None, None, None, :, None]
        elif reference_points.shape[-1] == 2:
            x = reference_points[:, :, None, :, None, 0] \
                + off / self.n_points * reference_points[:, :, None, :, None, 1] * 0.5
        else:
            raise ValueError('Last dim of reference_points must be 1 or 2, but get {} instead.'.format(
                reference_points.shape[-1]))
        return torch.stack((x, torch.full_like(x, 0.5)), -1)

    def _fused_eligible(self, query, input_flatten, host_lengths):
        """domain of the fused HIP path (include/gvl_msda.h: gvl_msda1d_fused_*)"""
        return (self.fused and host_lengths is not None and query.dtype in (torch.float32, torch.bfloat16)
                and self.d_model // self.n_heads == 64 and self.n_levels * self.n_points == 16 and self.n_points == 4
                and (input_flatten.shape[1] <= 600 or input_flatten.shape[1] - host_lengths[0][0] <= 600))

    def _cat_projection(self):
        """[sampling_offsets; attention_weights] weight and bias.  With autograd on, a differentiable cat (the gradient
        splits back onto the two layers); in inference kept per parameter version instead of rebuilt on every call."""
        ws = (self.sampling_offsets.weight, self.attention_weights.weight, self.sampling_offsets.bias,
              self.attention_weights.bias)
        if torch.is_grad_enabled() and any(p_.requires_grad for p_ in ws):
            return torch.cat(ws[:2], 0), torch.cat(ws[2:], 0)
        key = tuple((p_.data_ptr(), p_._version) for p_ in ws)
        cached = self.__dict__.get("_cat_proj")
        if cached is None or cached[0] != key:
            with torch.no_grad():
                cached = (key, torch.cat(ws[:2], 0), torch.cat(ws[2:], 0))
            self.__dict__["_cat_proj"] = cached
        return cached[1], cached[2]

    def _forward_fused(self, query, reference_points, input_flatten, shapes2d, level_start_index, padding_mask):
        N, Len_in, _ = input_flatten.shape
        value = self.value_proj(input_flatten)
        if padding_mask is not None:
            value = _mask_rows(value, padding_mask)
        value = value.view(N, Len_in, self.n_heads, self.d_model // self.n_heads)
        # one GEMM for both projections: columns [0,128) raw offsets, [128,256) attention logits
        q = query.contiguous()
        ws = (self.sampling_offsets.weight, self.attention_weights.weight)
        bs = (self.sampling_offsets.bias, self.attention_weights.bias)
        if train_linear_eligible(q, ws, bs):
            # training: both projections as ONE hand-written product over the stacked weight (no torch.cat of the parameters:
            # the planes of the stack come from the model's TrainPlanes), dx / dW / db from the hand-written backward kernels
            proj = train_linear(q, ws, bs, getattr(self, "defer_wgrad", False))
        else:
            w_cat, b_cat = self._cat_projection()
            proj = projection(q, w_cat, b_cat)
        out, amax = _fn.MSDeformAttnFusedFunction.apply(value, proj, reference_points.contiguous(), shapes2d,
                                                        level_start_index, self.n_levels, self.n_points, self.pad_mode,
                                                        torch.is_grad_enabled() and train_linear_eligible(
                                                            query, (self.output_proj.weight,), (self.output_proj.bias,)))
        if amax is not None:
            from ... import layers as _layers
            _layers.tag_amax(out, amax)
        return self.output_proj(out)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        """query (N, Lq, C); reference_points (N, Lq, L, 1|2); input_flatten (N, sum T_l, C);
        input_spatial_shapes (L,) = [T_0..]; input_level_start_index (L,); input_padding_mask (N, sum T_l) True=pad
        -> (N, Lq, C)   (ms_deform_attn.py:79-126)"""
        if query.device.type != 'cuda':
            raise RuntimeError("gvl_amd.MSDeformAttn runs on a ROCm device only (no CPU fallback); got "
                               f"{query.device}")
        host = getattr(input_spatial_shapes, "_gvl_host_lengths", None)
        if host is not None:                      # lengths known on the host: no device->host sync
            assert sum(host[0]) == input_flatten.shape[1]
        else:                                     # ms_deform_attn.py:93 (synchronises, as the reference does)
            assert input_spatial_shapes.sum() == input_flatten.shape[1]
        if reference_points.shape[-1] not in (1, 2):
            raise ValueError('Last dim of reference_points must be 1 or 2, but get {} instead.'.format(
                reference_points.shape[-1]))
        shapes2d = temporal_shapes_2d(input_spatial_shapes, input_level_start_index)
        if self._fused_eligible(query, input_flatten, host):
            return self._forward_fused(query, reference_points, input_flatten, shapes2d, input_level_start_index,
                                       input_padding_mask)
        value, off, aw = self._project(query, input_flatten, input_padding_mask)
        loc = self._locations(reference_points, off, input_spatial_shapes)
        if self.pad_mode == "zeros":
            out = _fn.MSDeformAttnFunction.apply(value, shapes2d, input_level_start_index, loc, aw, self.im2col_step)
        else:
            out = _fn.MSDeformAttnPadFunction.apply(value, shapes2d, input_level_start_index, loc, aw,
                                                    self.im2col_step, self.pad_mode)
        return self.output_proj(out)
