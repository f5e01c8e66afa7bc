"""Feature pyramid + positional encoding that feed the path -- mirror of pdvc/base_encoder.py:55-82 and
pdvc/position_encoding.py:38-64.  Plain PyTorch-ROCm ops (conv1d / GroupNorm): SURVEY.md section 8 keeps these out
of hand-kernel scope (row f2 "next").  Parameter names: ``input_proj.{l}.{0,1}``, ``pos_embed.duration_embed_layer``.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn


class _SinePosFunction(torch.autograd.Function):
    """PositionEmbeddingSine.forward of one level as ONE launch (include/gvl_msda.h: gvl_pos_embed_sine_f32) ->
    (N, n_sine + n_dur, T); only the duration embedding carries gradient (sum over time)."""

    @staticmethod
    def forward(ctx, mask, dim_t, dur, scale):
        from . import _lib
        N, T = mask.shape
        F_, Cd = dim_t.numel(), dur.shape[1]
        mask_u8 = mask.contiguous().view(torch.uint8)
        dur = dur.contiguous()
        out = torch.empty((N, F_ + Cd, T), dtype=torch.float32, device=mask.device)
        with torch.cuda.device(mask.device):
            rc = _lib.lib().gvl_pos_embed_sine_f32(mask_u8.data_ptr(), dim_t.data_ptr(), dur.data_ptr(), N, T, F_, Cd,
                                                   float(scale), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "pos_embed_sine")
        ctx.n_sine = F_
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return None, None, grad_out[:, ctx.n_sine:].sum(-1), None


class PositionEmbeddingSine(nn.Module):
    """sine over the (normalised) frame index ++ a learned embedding of the video duration (position_encoding.py)."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = 2 * math.pi if scale is None else scale
        self.max_duration = 256
        self.duration_embed_layer = nn.Linear(self.max_duration, self.max_duration)

    def duration_embedding(self, durations):
        """position_encoding.py:58-63 without the per-video Python loop: step function of the integer duration."""
        steps = torch.arange(self.max_duration, device=durations.device)[None, :]
        onehot = (steps < durations.int()[:, None]).to(self.duration_embed_layer.weight.dtype)
        return self.duration_embed_layer(onehot)

    def _dim_t(self, device):
        cached = getattr(self, "_dim_t_cache", None)
        if cached is None or cached.device != device:
            dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32, device=device)
            cached = self._dim_t_cache = (self.temperature ** (2 * (dim_t // 2) / self.num_pos_feats)).contiguous()
        return cached

    def forward(self, x, mask, duration, dur_embed=None):
        if mask.is_cuda and self.normalize and self.duration_embed_layer.weight.dtype == torch.float32 \
                and not torch.is_autocast_enabled():
            dur = self.duration_embedding(duration) if dur_embed is None else dur_embed
            return _SinePosFunction.apply(mask, self._dim_t(mask.device), dur, self.scale)
        not_mask = ~mask
        x_embed = not_mask.cumsum(1, dtype=torch.float32)
        if self.normalize:
            x_embed = (x_embed - 0.5) / (x_embed[:, -1:] + 1e-6) * self.scale
        dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32, device=x.device)
        dim_t = self.temperature ** (2 * (dim_t // 2) / self.num_pos_feats)
        pos_x = x_embed[:, :, None] / dim_t
        pos_x = torch.stack((pos_x[:, :, 0::2].sin(), pos_x[:, :, 1::2].cos()), dim=3).flatten(2)
        dur = (self.duration_embedding(duration) if dur_embed is None else dur_embed)
        dur = dur.reshape(-1, 1, self.max_duration).expand_as(pos_x)
        return torch.cat((pos_x, dur), dim=2).permute(0, 2, 1)


class BaseEncoder(nn.Module):
    def __init__(self, num_feature_levels, vf_dim, hidden_dim):
        super().__init__()
        self.pos_embed = PositionEmbeddingSine(hidden_dim // 2, normalize=True)
        self.num_feature_levels = num_feature_levels
        self.hidden_dim = hidden_dim
        if num_feature_levels > 1:
            projs = [nn.Sequential(nn.Conv1d(vf_dim, hidden_dim, kernel_size=1), nn.GroupNorm(32, hidden_dim))]
            in_ch = vf_dim
            for _ in range(num_feature_levels - 1):
                projs.append(nn.Sequential(nn.Conv1d(in_ch, hidden_dim, kernel_size=3, stride=2, padding=1),
                                           nn.GroupNorm(32, hidden_dim)))
                in_ch = hidden_dim
            self.input_proj = nn.ModuleList(projs)
        else:
            self.input_proj = nn.ModuleList([nn.Sequential(nn.Conv2d(vf_dim, hidden_dim, kernel_size=1),
                                                           nn.GroupNorm(32, hidden_dim))])
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)

    @staticmethod
    def _conv_as_gemm(seq, x):
        """Conv1d(k=1) / Conv1d(k=3, stride 2, pad 1) + GroupNorm of one pyramid level with the convolution written
        as one GEMM over gathered taps: MIOpen's solver choice for these tiny 1-D convolutions on gfx950 is a naive
        direct kernel (0.7 ms per call in the rocprof trace), a plain GEMM is ~50x faster.  Same arithmetic."""
        conv, norm = seq[0], seq[1]
        w = conv.weight
        if conv.kernel_size[0] == 1:
            y = torch.matmul(w[:, :, 0], x)
        else:
            T = x.shape[-1]
            t_out = (T - 1) // 2 + 1
            xp = F.pad(x, (1, 1))
            cols = torch.cat([xp[:, :, k:k + 2 * t_out:2] for k in range(3)], dim=1)      # (N, 3*C_in, T_out)
            y = torch.matmul(w.permute(0, 2, 1).reshape(w.shape[0], -1), cols)
        return norm(y + conv.bias[None, :, None])

    def forward(self, vf, mask, duration):
        """vf (N,T,C_in), mask (N,T) True=pad, duration (N,) -> lists over levels of (N,C,T_l), (N,T_l), (N,C,T_l)"""
        assert mask is not None
        x = vf.transpose(1, 2)
        if self.num_feature_levels == 1:
            return [self.input_proj[0](x)], [mask], [self.pos_embed(x, mask, duration)]
        srcs = [self._conv_as_gemm(self.input_proj[0], x)]
        masks = [mask]
        dur = self.pos_embed.duration_embedding(duration)         # the same for every level: computed once
        poses = [self.pos_embed(x, mask, duration, dur_embed=dur)]
        for l in range(1, self.num_feature_levels):
            src = self._conv_as_gemm(self.input_proj[l], x if l == 1 else srcs[-1])
            m = F.interpolate(mask[None].float(), size=src.shape[-1:]).to(torch.bool)[0]
            srcs.append(src)
            masks.append(m)
            poses.append(self.pos_embed(src, m, duration, dur_embed=dur).to(src.dtype))
        return srcs, masks, poses


def build_base_encoder(args):
    return BaseEncoder(args.num_feature_levels, args.feature_dim, args.hidden_dim)
