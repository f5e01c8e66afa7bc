"""MSDeformAttnCap -- mirror of pdvc/ops/modules/ms_deform_attn_for_caption.py:30-127 on the HIP sampler.

The captioner's variant takes 2C (or 3C) wide queries, centres the offset bias over the points (:72), and returns
the *unweighted* sampled values (B*M, D, Lq, L, P) with border padding -- the reference calls the PyTorch core with
return_value=True even on the GPU (:124-125).  ``attention_weights`` / ``output_proj`` exist (checkpoint
compatibility) but do not influence the output, exactly as in the reference.
"""
import warnings

import torch
from torch import nn

from .ms_deform_attn import _mask_rows, _power_of_two, offset_bias_init, temporal_shapes_2d
from ... import MultiScaleDeformableAttention as MSDA
from ...linear import Linear


class MSDeformAttnCap(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, opt=None):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError('d_model must be divisible by n_heads, but got {} and {}'.format(d_model, n_heads))
        if not _power_of_two(d_model // n_heads):
            warnings.warn("You'd better set d_model in MSDeformAttn to make the dimension of each attention head a "
                          "power of 2 which is more efficient in our CUDA implementation.")
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        qdim = (3 if (opt is not None and vars(opt).get('enable_pos_emb_for_captioner')) else 2) * d_model
        self.sampling_offsets = nn.Linear(qdim, n_heads * n_levels * n_points)
        self.attention_weights = nn.Linear(qdim, n_heads * n_levels * n_points)
        # (gvl_amd Linear: in training at >= 512 rows the hand-written product on the model's operand planes, its weight gradient
        #  in the backward pass's grouped launch)
        self.value_proj = Linear(d_model, d_model)
        self.value_proj.defer_wgrad = True
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        with torch.no_grad():
            self.sampling_offsets.weight.zero_()
            self.sampling_offsets.bias.copy_(offset_bias_init(self.n_heads, self.n_levels, self.n_points, centre=True))
            self.attention_weights.weight.zero_()
            self.attention_weights.bias.zero_()
            nn.init.xavier_uniform_(self.value_proj.weight)
            self.value_proj.bias.zero_()
            nn.init.xavier_uniform_(self.output_proj.weight)
            self.output_proj.bias.zero_()

    def project_value(self, input_flatten, input_padding_mask=None):
        """value_proj(memory) with padded rows zeroed (:98-101).  Exposed so a token loop can hoist it: the
        reference recomputes this identical tensor at every decoding step."""
        N, Len_in, _ = input_flatten.shape
        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = _mask_rows(value, input_padding_mask)
        return value.view(N, Len_in, self.n_heads, self.d_model // self.n_heads)

    def sampling_locations(self, query, reference_points, input_spatial_shapes):
        N, Len_q, _ = query.shape
        off = self.sampling_offsets(query).view(N, Len_q, self.n_heads, self.n_levels, self.n_points)
        if reference_points.shape[-1] == 1:
            x = reference_points[:, :, None, :, None, 0] + off / input_spatial_shapes[None, None, None, :, None]
        elif reference_points.shape[-1] == 2:
            x = reference_points[:, :, None, :, None, 0] \
                + off / self.n_points * reference_points[:, :, None, :, None, 1] * 0.5
        else:
            raise ValueError('Last dim of reference_points must be 1 or 2, but get {} instead.'.format(
                reference_points.shape[-1]))
        return torch.stack((x, torch.full_like(x, 0.5)), -1)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, value=None):
        if query.device.type != 'cuda':
            raise RuntimeError("gvl_amd.MSDeformAttnCap runs on a ROCm device only (no CPU fallback); got "
                               f"{query.device}")
        if value is None:
            value = self.project_value(input_flatten, input_padding_mask)
        loc = self.sampling_locations(query, reference_points, input_spatial_shapes)
        shapes2d = temporal_shapes_2d(input_spatial_shapes, input_level_start_index)
        return MSDA.ms_deform_attn_sample(value.contiguous(), shapes2d, input_level_start_index, loc.contiguous(),
                                          "border")
