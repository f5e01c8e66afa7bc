"""Feature pyramid + positional encoding that feed the path -- mirror of pdvc/base_encoder.py:55-82 and
pdvc/position_encoding.py:38-64.  Plain PyTorch-ROCm ops (conv1d / GroupNorm): SURVEY.md section 8 keeps these out
of hand-kernel scope (row f2 "next").  Parameter names: ``input_proj.{l}.{0,1}``, ``pos_embed.duration_embed_layer``.
"""
import ctypes
import os
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
        hit = getattr(grad_out, "_gvl_time_sum", None)       # (N, C) sums over time left by gvl_amd.layers._LevelPosEmbed's backward
        if hit is not None and hit[1] == grad_out._version and hit[0].shape == grad_out.shape[:2]:
            return None, None, hit[0][:, ctx.n_sine:], None
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


def _keeps_products(cls):
    from . import MultiScaleDeformableAttention as MSDA                  # (imported late, as everywhere in this module)
    return MSDA.keeps_products(cls)


@_keeps_products
class _PyramidTrainFunction(torch.autograd.Function):
    """TRAINING form of the feature pyramid (base_encoder.py:60-80): Conv1d(k = 1) / Conv1d(k = 3, stride 2, padding 1) + GroupNorm of
    every level -> the levels' rows of the flattened (N, S, C) encoder input, on the kernels of the inference form (forward_flat)
    plus their gradients:
        conv          gvl_linear_f16x3_f32 over rows of taps (one copy gathers the three taps of a frame: the weight gradient's operand)
        GroupNorm     gvl_group_norm_rows_f32 / gvl_group_norm_rows_backward_f32 (the backward also takes the next level's input
                      gradient, in its padded layout)
        dW, db        gvl_wgrad_f16x3_f32;   dx (levels >= 2)  gvl_linear_f16x3_f32 on the transposed planes + gvl_conv_taps_to_rows_f32
    ~60 launches forward + backward where the PyTorch formulation (pad, unfold copy, matmul, bias add, native group norm and their
    autograd) takes ~140.  params: (conv.weight, conv.bias, norm.weight, norm.bias) per level."""

    @staticmethod
    def forward(ctx, enc, vf, *params):
        from . import _lib
        from . import layers as L
        from . import linear as GL
        from .train_planes import Operand
        N, T, Cin = vf.shape
        C, nl = enc.hidden_dim, enc.num_feature_levels
        lengths = enc.level_lengths(T)
        starts = [sum(lengths[:i]) for i in range(nl)]
        S = sum(lengths)
        dev = vf.device
        stream = torch.cuda.current_stream().cuda_stream
        tp, mats = enc._train_conv_operands(params)
        src = torch.empty(N, S, C, device=dev, dtype=torch.float32)
        saved, shapes = [], []
        x = vf.reshape(N * T, Cin)
        x = x if x.is_contiguous() else x.contiguous()
        xp = None
        for l in range(nl):
            if l == 0:
                a, rows = x, T
            else:
                t_out = lengths[l]
                rows = t_out + 1
                a = xp.as_strided((N * rows, 3 * xp.shape[2]), (2 * xp.shape[2], 1)).contiguous()     # rows of taps
            am = L.row_absmax(a)[0]
            fwd, _, bias = tp.lookup((mats[l],))
            y = torch.empty(a.shape[0], C, device=dev, dtype=torch.float32)
            op = Operand(fwd, C, a.shape[1], bias)
            if (l > 0 and a.shape[0] <= 1024 and op.K <= 4096 and L.splitk_pays(a.shape[0], op.N, op.K)
                    and os.environ.get("GVL_CONV_SPLITK", "1") != "0"):
                L.linear_splitk(a, am, op, out=y, bias=True)          # (few rows of taps, contraction 3 C_in: as forward_flat)
            else:
                L.linear(a, op, [L.seg(0, y, am)])
            norm_w, norm_b = params[4 * l + 2], params[4 * l + 3]
            nxt = None
            if l + 1 < nl:                                             # next level's zero-padded input: (N + 1, 2 (T'' + 1), C)
                t_nxt = lengths[l + 1]
                nxt = torch.zeros(N + 1, 2 * (t_nxt + 1), C if l > 0 else Cin, device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                rc = _lib.lib().gvl_group_norm_rows_f32(
                    y.data_ptr(), y.stride(0), rows, N, lengths[l], C, enc.input_proj[l][1].num_groups, norm_w.data_ptr(),
                    norm_b.data_ptr(), float(enc.input_proj[l][1].eps), src.data_ptr() + 4 * starts[l] * C, S * C,
                    nxt.data_ptr() + 4 * C if (nxt is not None and l > 0) else None,
                    nxt.shape[1] * C if (nxt is not None and l > 0) else 0, stream)
            _lib.check(rc, "group_norm_rows")
            if l == 0 and nxt is not None:
                nxt[:N, 1:T + 1] = vf                                  # level 1 convolves the RAW features
            saved += [a, am, y]
            shapes.append(rows)
            xp = nxt
        ctx.save_for_backward(*saved, *params)
        ctx.enc, ctx.geom, ctx.tp, ctx.mats = enc, (N, T, Cin, C, nl, lengths, starts, S, shapes), tp, mats
        return src

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dsrc):
        from . import _lib
        from . import layers as L
        from . import MultiScaleDeformableAttention as MSDA
        from . import train_layers as TL
        from .train_planes import Operand
        enc, tp, mats = ctx.enc, ctx.tp, ctx.mats
        N, T, Cin, C, nl, lengths, starts, S, shapes = ctx.geom
        t_ = ctx.saved_tensors
        saved, params = t_[:3 * nl], t_[3 * nl:]
        dev = dsrc.device
        stream = torch.cuda.current_stream().cuda_stream
        dsrc = dsrc.contiguous()
        grads = [None] * (4 * nl)
        dnext = None                                                   # input gradient of level l + 1, padded layout (N, 2 T1, C)
        part_all = torch.empty(nl, 2, N, C, device=dev, dtype=torch.float32)        # per-video d gamma | d beta of every level
        for l in range(nl - 1, -1, -1):
            a, am, y = saved[3 * l], saved[3 * l + 1], saved[3 * l + 2]
            rows = shapes[l]
            norm = enc.input_proj[l][1]
            dy = torch.empty(N * rows, C, device=dev, dtype=torch.float32)
            am_dy = TL.step_zeros(N * rows, dev)                       # max |dy row|: left by the group norm's backward kernel
            part = part_all[l]
            with torch.cuda.device(dev):
                rc = _lib.lib().gvl_group_norm_rows_backward_amax_f32(
                    y.data_ptr(), y.stride(0), rows, N, lengths[l], C, norm.num_groups, params[4 * l + 2].data_ptr(), float(norm.eps),
                    dsrc.data_ptr() + 4 * starts[l] * C, S * C,
                    dnext.data_ptr() + 4 * C if dnext is not None else None, dnext.shape[1] * C if dnext is not None else 0,
                    dy.data_ptr(), C, part[0].data_ptr(), part[1].data_ptr(), am_dy.data_ptr(), stream)
            _lib.check(rc, "group_norm_rows_backward")
            gw, gbias = MSDA.wgrad(dy, a, am_dy, am)
            w = params[4 * l]
            grads[4 * l] = gw.view(C, 1, a.shape[1]).permute(0, 2, 1) if l == 0 else gw.view(C, 3, a.shape[1] // 3).permute(0, 2, 1)
            grads[4 * l + 1] = gbias
            dnext = None
            if l >= 2:                                                 # (level 1 reads the raw features: no input gradient)
                _, tr, _ = tp.lookup((mats[l],))
                dcols = torch.empty(N * rows, a.shape[1], device=dev, dtype=torch.float32)
                L.linear(dy, Operand(tr, a.shape[1], C, None), [L.seg(0, dcols, am_dy)])
                dnext = torch.empty(N, 2 * rows, C, device=dev, dtype=torch.float32)
                with torch.cuda.device(dev):
                    rc = _lib.lib().gvl_conv_taps_to_rows_f32(dcols.data_ptr(), N, rows, C, dnext.data_ptr(), stream)
                _lib.check(rc, "conv_taps_to_rows")
            assert w.shape[0] == C
        gb = part_all.sum(2)                                           # (nl, 2, C): the videos in index order, ONE reduction
        for l in range(nl):
            grads[4 * l + 2], grads[4 * l + 3] = gb[l, 0], gb[l, 1]
        return (None, None, *grads)


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
            # taps k = 0, 1, 2 of every output frame as one strided view (unfold) and one copy: (N, 3*C_in, T_out) with
            # row k * C_in + c -- what cat([xp[:, :, k::2]]) builds from three slices, whose backward is three zero-filled
            # strided SliceBackward + two accumulations per level instead of one UnfoldBackward
            cols = xp.unfold(2, 3, 2)[:, :, :t_out].permute(0, 3, 1, 2).reshape(x.shape[0], -1, t_out)
            y = torch.matmul(w.permute(0, 2, 1).reshape(w.shape[0], -1), cols)
        return norm(y + conv.bias[None, :, None])

    # -- inference: the whole pyramid flattened, on the hand-written kernels ------------------------------------------
    def flat_eligible(self, vf, mask):
        from . import layers as L
        c0 = self.input_proj[0][0]
        return (L.enabled() and not torch.is_grad_enabled() and not self.training and not torch.is_autocast_enabled()
                and vf.is_cuda and vf.dtype == torch.float32 and 1 < self.num_feature_levels <= 8
                and isinstance(c0, nn.Conv1d) and vf.shape[-1] % 32 == 0 and self.hidden_dim % 64 == 0
                and self.pos_embed.normalize and self.pos_embed.num_pos_feats + self.pos_embed.max_duration == self.hidden_dim
                and all(64 % (self.hidden_dim // p_[1].num_groups) == 0 for p_ in self.input_proj)
                and vf.shape[1] >= 2 ** (self.num_feature_levels - 1))

    def level_lengths(self, T):
        lengths = [T]
        for _ in range(self.num_feature_levels - 1):
            lengths.append((lengths[-1] - 1) // 2 + 1)
        return lengths

    def flat_train_eligible(self, vf, mask):
        """the TRAINING forward of the pyramid on the hand-written kernels (_PyramidTrainFunction); GVL_TRAIN_PYRAMID=torch keeps
        the PyTorch formulation (A/B switch)"""
        import os
        from . import layers as L
        from . import linear as GL
        c0 = self.input_proj[0][0]
        return (os.environ.get("GVL_TRAIN_PYRAMID", "") != "torch" and L.enabled() and GL.train_linear_enabled()
                and torch.is_grad_enabled() and self.training and not torch.is_autocast_enabled()
                and vf.is_cuda and vf.dtype == torch.float32 and 1 < self.num_feature_levels <= 8
                and isinstance(c0, nn.Conv1d) and vf.shape[-1] % 64 == 0 and self.hidden_dim % 64 == 0
                and all(64 % (self.hidden_dim // p_[1].num_groups) == 0 for p_ in self.input_proj)
                and all(p_[0].weight.dtype == torch.float32 for p_ in self.input_proj)
                and vf.shape[1] >= 2 ** (self.num_feature_levels - 1) and vf.shape[0] * vf.shape[1] >= 64)

    def _train_conv_operands(self, params):
        """the levels' convolutions as (C_out, k C_in) matrices in persistent buffers (row c_out: tap-major, as the rows of taps),
        refreshed from the parameters' current values, and their operand planes in both orientations (a private TrainPlanes:
        two launches for all levels)"""
        from .train_planes import TrainPlanes
        st = self.__dict__.get("_gvl_train_conv")
        dev = params[0].device
        if st is None or st[0].device != torch.device(dev):
            mats = []
            for l in range(self.num_feature_levels):
                w = params[4 * l]
                mats.append(torch.empty(w.shape[0], w.shape[1] * w.shape[2], device=dev, dtype=torch.float32))
            tp = TrainPlanes(dev)
            for l, m in enumerate(mats):
                tp.register([m], [params[4 * l + 1]])
            st = self.__dict__["_gvl_train_conv"] = (tp, mats)
        tp, mats = st
        with torch.no_grad():
            for l, m in enumerate(mats):
                w = params[4 * l]
                m.view(w.shape[0], w.shape[2], w.shape[1]).copy_(w.permute(0, 2, 1))
            tp.refresh()
        return tp, mats

    def forward_flat_train(self, vf):
        """-> the flattened (N, S, C) encoder input with gradients to the levels' conv / norm parameters"""
        params = []
        for seq in self.input_proj:
            params += [seq[0].weight, seq[0].bias, seq[1].weight, seq[1].bias]
        return _PyramidTrainFunction.apply(self, vf, *params)

    def train_geometry(self, vf, mask, duration):
        """the levels' padding masks and position embeddings of forward() without the pyramid itself -> (masks, poses)"""
        lengths = self.level_lengths(vf.shape[1])
        dur = self.pos_embed.duration_embedding(duration)
        masks, poses = [mask], [self.pos_embed(None, mask, duration, dur_embed=dur)]
        for l in range(1, self.num_feature_levels):
            m = F.interpolate(mask[None].float(), size=(lengths[l],)).to(torch.bool)[0]
            masks.append(m)
            poses.append(self.pos_embed(None, m, duration, dur_embed=dur))
        return masks, poses

    def _conv_weights(self, l):
        """the level's convolution as the (C_out, k * C_in) matrix of a product over rows of taps, as split planes"""
        from . import layers as L
        conv = self.input_proj[l][0]
        key = (conv.weight.data_ptr(), conv.weight._version, conv.bias._version)
        hit = conv.__dict__.get("_gvl_conv_w")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                w = conv.weight.detach()
                w2 = w[:, :, 0] if conv.kernel_size[0] == 1 else w.permute(0, 2, 1).reshape(w.shape[0], -1)
                hit = conv.__dict__["_gvl_conv_w"] = (key, L.Weights([(w2.contiguous(), conv.bias.detach())]))
        return hit[1]

    def forward_flat(self, vf, mask, duration, level_embed):
        """Inference form of forward() + DeformableTransformer.prepare_encoder_inputs (deformable_transformer.py:85-115):
        -> (src_flatten (N, S, C), mask_flatten (N, S) bool, lvl_pos_embed_flatten (N, S, C), level lengths).
        Every conv1d is one gvl_linear_f16x3_f32 product -- the k = 3, stride 2 levels read their three taps as ONE row of
        a strided view of the zero-padded input ((N, 2 (T' + 1), C_in) rows: tap row 2 t' - 1 .. 2 t' + 1 of frame t' are
        contiguous) --, GroupNorm writes each level straight into its rows of the flattened tensor (and into the next
        level's padded input), one launch produces all masks / position / level embeddings (gvl_pyramid_geometry_f32)."""
        from . import _lib
        from . import layers as L
        N, T, Cin = vf.shape
        C, nl = self.hidden_dim, self.num_feature_levels
        lengths = [T]
        for _ in range(nl - 1):
            lengths.append((lengths[-1] - 1) // 2 + 1)
        starts = [sum(lengths[:i]) for i in range(nl)]
        S = sum(lengths)
        dev = vf.device
        src = torch.empty(N, S, C, device=dev, dtype=torch.float32)
        stream = torch.cuda.current_stream().cuda_stream

        def group_norm(y, rows_per_video, l, dst2=None, dst2_vs=0):
            norm = self.input_proj[l][1]
            with torch.cuda.device(dev):
                rc = _lib.lib().gvl_group_norm_rows_f32(
                    y.data_ptr(), y.stride(0), rows_per_video, N, lengths[l], C, norm.num_groups, norm.weight.data_ptr(),
                    norm.bias.data_ptr(), float(norm.eps), src.data_ptr() + 4 * starts[l] * C, S * C,
                    dst2.data_ptr() + 4 * C if dst2 is not None else None, dst2_vs, stream)
            _lib.check(rc, "group_norm_rows")

        # the zero-padded inputs of the strided levels: ONE fill for all of them
        pad_in = [T] + [lengths[l] for l in range(1, nl - 1)]                  # frames going into the convolution of level 1, 2, ..
        pad_ch = [Cin] + [C] * (nl - 2)
        pad_n = [(N + 1) * 2 * ((t_ - 1) // 2 + 2) * c_ for t_, c_ in zip(pad_in, pad_ch)]
        pad_pool = torch.zeros(sum(pad_n), device=dev, dtype=torch.float32) if nl > 1 else None
        pad_next = [0, 0]                                                      # (buffers handed out, elements handed out)

        def padded(t_in, ch):
            """zeroed (N + 1, 2 (T' + 1), ch) buffer: row 0 and the rows behind the t_in frames are the convolution's padding
            (and slack for the last, unused tap row of every video); the extra video keeps the strided view in bounds"""
            t_out = (t_in - 1) // 2 + 1
            k, at = pad_next
            assert (t_in, ch) == (pad_in[k], pad_ch[k])
            pad_next[0], pad_next[1] = k + 1, at + pad_n[k]
            return pad_pool[at:at + pad_n[k]].view(N + 1, 2 * (t_out + 1), ch), t_out

        def conv_s2(xp, t_out, ch, l):
            a = xp.as_strided((N * (t_out + 1), 3 * ch), (2 * ch, 1))           # row (n, t'): taps 2 t' - 1 .. 2 t' + 1
            am, _ = L.row_absmax(a)
            y = torch.empty(N * (t_out + 1), C, device=dev, dtype=torch.float32)
            w = self._conv_weights(l)
            if (a.shape[0] <= 1024 and w.K <= 4096 and L.splitk_pays(a.shape[0], w.N, w.K)
                    and os.environ.get("GVL_CONV_SPLITK", "1") != "0"):
                # a few hundred rows of taps against a contraction of 3 C_in: 13-52 tiles of 48 K stages -- split over K.  (Not for
                # thousands of rows per level -- long videos: their tiles fill the chip as they are -- nor for the 9216-long
                # contraction of 3072-d features: the T = 512 parity test sits within 2 % of its tolerance and the partial sums'
                # different rounding moves it across)
                L.linear_splitk(a, am, w, out=y, bias=True)
            else:
                L.linear(a, w, [L.seg(0, y, am)])
            return y

        x = vf.reshape(N * T, Cin)
        if not x.is_contiguous():
            x = x.contiguous()
        am, _ = L.row_absmax(x)
        y0 = torch.empty(N * T, C, device=dev, dtype=torch.float32)
        L.linear(x, self._conv_weights(0), [L.seg(0, y0, am)])
        group_norm(y0, T, 0)
        xp, t_out = padded(T, Cin)                                             # level 1 convolves the RAW features
        xp[:N, 1:T + 1] = vf
        ch = Cin
        for l in range(1, nl):
            y = conv_s2(xp, t_out, ch, l)
            nxt = padded(lengths[l], C) if l + 1 < nl else (None, 0)
            group_norm(y, t_out + 1, l, nxt[0], nxt[0].shape[1] * C if nxt[0] is not None else 0)
            xp, t_out, ch = nxt[0], nxt[1], C
        dur = self.pos_embed.duration_embedding(duration).contiguous()
        mask_u8 = mask.contiguous().view(torch.uint8)
        mask_flat = torch.empty(N, S, device=dev, dtype=torch.uint8)
        lvl_pos = torch.empty(N, S, C, device=dev, dtype=torch.float32)
        arr = (ctypes.c_int64 * nl)
        pe = self.pos_embed
        with torch.cuda.device(dev):
            rc = _lib.lib().gvl_pyramid_geometry_f32(
                mask_u8.data_ptr(), N, T, S, nl, arr(*lengths), arr(*starts), pe._dim_t(dev).data_ptr(), dur.data_ptr(),
                level_embed.contiguous().data_ptr(), pe.num_pos_feats, pe.max_duration, float(pe.scale),
                mask_flat.data_ptr(), lvl_pos.data_ptr(), stream)
        _lib.check(rc, "pyramid_geometry")
        return src, mask_flat.view(torch.bool), lvl_pos, lengths

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
