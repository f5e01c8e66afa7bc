"""Inference form of the transformer layers around the deformable attention: every ``nn.Linear`` of the encoder /
decoder layers (pdvc/deformable_transformer.py:189-199,257-280; pdvc/ops/modules/ms_deform_attn.py:95,99-100,125) and
the box MLP (pdvc/pdvc.py:1166-1178) on ``gvl_linear_f16x3_f32`` -- fp32 activations split into fp16 planes inside the
kernel's load path, three fp16 MFMAs per product, fp32 accuracy (include/gvl_msda.h) -- with bias / ReLU / residual /
masked rows in the GEMM's epilogue and LayerNorm as one kernel that also leaves the row maxima the next product needs.

The modules keep their parameters, names and training path; ``DeformableTransformerEncoder / Decoder.forward`` call
into this file when ``enabled()`` and the call is inference (no grad, eval mode, fp32, shapes in the kernels' domain).
``GVL_LAYERS=torch`` keeps the PyTorch formulation for A/B runs.

Row maxima ("amax"): the split needs a power-of-two scale per activation row.  Every producer on the path leaves
max |row| behind for its consumer -- LayerNorm (register-local), the sampling kernel and the GEMM epilogue (one atomic
max per row and tile into a zero-initialised vector), ``row_absmax`` for tensors that come from PyTorch ops.
"""
import ctypes
import os

import torch

from . import MultiScaleDeformableAttention as MSDA
from . import _lib

LIN_ADDEND, LIN_RELU = 1, 2
LIN_XCD_COLUMNS = 1


def enabled():
    return os.environ.get("GVL_LAYERS", "") != "torch"


class _Seg(ctypes.Structure):                      # include/gvl_msda.h: gvl_lin_seg
    _fields_ = [("n_begin", ctypes.c_int), ("flags", ctypes.c_int), ("out", ctypes.c_void_p), ("ldo", ctypes.c_int64),
                ("amax_in", ctypes.c_void_p), ("resid", ctypes.c_void_p), ("ldr", ctypes.c_int64),
                ("amax_out", ctypes.c_void_p), ("rowmask", ctypes.c_void_p), ("width", ctypes.c_int)]


class Weights:
    """[W_0; W_1; ...] (each padded to a multiple of 64 rows) as fp16 planes + the concatenated bias; ``starts[i]`` is the
    first column of block i.  Built once per parameter version (``cached``)."""

    def __init__(self, pairs, pad_to=64):
        """pad_to=128: every block starts at a multiple of 128 columns, which lets the kernel use its 128 x 128 tile"""
        ws, bs, self.starts, self.widths = [], [], [], []
        n = 0
        for w, b in pairs:
            w = w.detach()
            rows = w.shape[0]
            pad = (-rows) % pad_to
            self.starts.append(n)
            self.widths.append(rows)
            ws.append(w if not pad else torch.cat([w, w.new_zeros(pad, w.shape[1])], 0))
            bias = b.detach() if b is not None else w.new_zeros(rows)
            bs.append(bias if not pad else torch.cat([bias, bias.new_zeros(pad)], 0))
            n += rows + pad
        self.N, self.K = n, ws[0].shape[1]
        with torch.no_grad(), torch.autocast("cuda", enabled=False):
            self.planes = MSDA.split_rows(torch.cat(ws, 0).float().contiguous())
            self.bias = torch.cat(bs, 0).float().contiguous()


def cached(owner, name, params, pad_to=None, key_of=None):
    """Weights of `params` = [(weight, bias), ...] kept on `owner` until one of the parameters changes.  Several blocks
    are padded to multiples of 128 columns when that keeps the launch on the wide tile (total >= 256 columns).
    key_of: the PARAMETERS the blocks derive from, with `params` a callable that builds the blocks on a miss -- for blocks
    that are not views of a parameter (a row permutation: a new tensor, and a new address, on every evaluation)."""
    if key_of is not None:
        key = tuple((p_.data_ptr(), p_._version) for p_ in key_of)
    else:
        key = tuple((p_.data_ptr(), p_._version) for pair in params for p_ in pair if p_ is not None)
    store = owner.__dict__.setdefault("_gvl_lin_w", {})
    hit = store.get(name)
    if hit is None or hit[0] != key:
        if callable(params):
            params = params()
        if pad_to is None:
            pad_to = 128 if len(params) > 1 and sum(w_.shape[0] for w_, _ in params) >= 256 else 64
        hit = store[name] = (key, Weights(params, pad_to))
    return hit[1]


def seg(n_begin, out, amax_in, resid=None, amax_out=None, rowmask=None, relu=False, addend=False, width=0):
    """one column segment of a linear() launch; width > 0: only the first `width` columns of the segment are stored (a
    weight block padded to a multiple of 64 rows)"""
    return dict(n_begin=n_begin, out=out, amax_in=amax_in, resid=resid, amax_out=amax_out, rowmask=rowmask, relu=relu,
                addend=addend, width=width)


def linear(a, w, segs, a2=None, flags=0):
    """gvl_linear_f16x3_f32: a (R, K) fp32 (row stride a.stride(0)); a2 (rows2, K) the addend of ADDEND segments, row
    r uses a2[r % rows2]; w: Weights; segs: list of seg(...) -- the outputs are written in place."""
    R, K = a.shape
    assert a.dtype == torch.float32 and a.stride(1) == 1 and K == w.K, (a.shape, a.stride(), w.K)
    arr = (_Seg * len(segs))()
    for i, s in enumerate(segs):
        o = s["out"]
        assert o.dtype == torch.float32 and o.stride(1) == 1 and o.shape[0] == R
        r_ = s["resid"]
        assert s["amax_in"].numel() == R and s["amax_in"].dtype == torch.float32
        arr[i] = _Seg(s["n_begin"], (LIN_ADDEND if s["addend"] else 0) | (LIN_RELU if s["relu"] else 0), o.data_ptr(),
                      o.stride(0), s["amax_in"].data_ptr(), r_.data_ptr() if r_ is not None else None,
                      r_.stride(0) if r_ is not None else 0,
                      s["amax_out"].data_ptr() if s["amax_out"] is not None else None,
                      s["rowmask"].data_ptr() if s["rowmask"] is not None else None, s["width"])
    with torch.cuda.device(a.device):
        rc = _lib.lib().gvl_linear_f16x3_f32(
            a.data_ptr(), a.stride(0), a2.data_ptr() if a2 is not None else None, a2.stride(0) if a2 is not None else 0,
            a2.shape[0] if a2 is not None else 0, R, K, w.planes.hi.data_ptr(), w.planes.lo.data_ptr(),
            w.planes.scale.data_ptr(), w.bias.data_ptr() if w.bias is not None else None, w.N, arr, len(segs), flags,
            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "linear_f16x3")


_SPLITK_WS = {}


def _splitk_workspace(n_floats, device):
    """the partial slabs of a split-K product: ONE buffer per device, grown on demand and kept (successive products on a stream
    use it in turn; nothing else ever aliases it).  Not during a stream capture: a captured product gets a buffer of the graph's
    own pool, as every other temporary of the capture."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(n_floats, device=device, dtype=torch.float32)
    key = str(device)
    ws = _SPLITK_WS.get(key)
    if ws is None or ws.numel() < n_floats:
        ws = _SPLITK_WS[key] = torch.empty(n_floats, device=device, dtype=torch.float32)
    return ws


def linear_splitk(a, amax_a, w, out=None, bias=False):
    """gvl_linear_f16x3_splitk_bias_f32: out (R, N) = a (R, K) . w^T (+ w.bias when bias=True) for few outputs and a long
    contraction (w: operand planes of N rows, N % 128 == 0; a's row stride may exceed K -- the padding must be finite and
    multiplies zero planes -- or, with overlapping rows, fall short of it)"""
    R = a.shape[0]
    K = w.K
    assert a.dtype == torch.float32 and a.stride(1) == 1 and a.stride(0) > 0 and amax_a.numel() == R
    if out is None:
        out = torch.empty(R, w.N, device=a.device, dtype=torch.float32)
    L_ = _lib.lib()
    nbytes = L_.gvl_linear_f16x3_splitk_workspace_bytes(R, w.N, K)
    ws = _splitk_workspace(max(nbytes, 16) // 4, a.device)
    b = w.bias if bias and w.bias is not None else None
    with torch.cuda.device(a.device):
        rc = L_.gvl_linear_f16x3_splitk_bias_f32(a.data_ptr(), a.stride(0), amax_a.data_ptr(), R, K, w.planes.hi.data_ptr(),
                                                 w.planes.lo.data_ptr(), w.planes.scale.data_ptr(),
                                                 b.data_ptr() if b is not None else None, w.N, out.data_ptr(), ws.data_ptr(),
                                                 nbytes, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "linear_f16x3_splitk")
    return out


def splitk_pays(R, N, K):
    """whether gvl_linear_f16x3_splitk_* would split this product at all (few tiles, a long contraction)"""
    return N % 128 == 0 and K % 32 == 0 and _lib.lib().gvl_linear_f16x3_splitk_workspace_bytes(R, N, K) > 0


def layer_norm(x, norm, pos=None, want_amax=True):
    """LayerNorm of the rows of x (R, C) -> (y, amax_y, amax_{y + pos} or None)   (gvl_layer_norm_rows_f32)"""
    R, C = x.shape
    y = torch.empty_like(x)
    am = torch.empty(R, device=x.device, dtype=torch.float32) if want_amax else None
    amp = torch.empty(R, device=x.device, dtype=torch.float32) if pos is not None else None
    with torch.cuda.device(x.device):
        rc = _lib.lib().gvl_layer_norm_rows_f32(
            x.data_ptr(), R, C, norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps),
            pos.data_ptr() if pos is not None else None, pos.shape[0] if pos is not None else 0, y.data_ptr(),
            am.data_ptr() if am is not None else None, amp.data_ptr() if amp is not None else None,
            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "layer_norm_rows")
    return y, am, amp


def row_absmax(x, pos=None, want_x=True):
    """(max |x[r]|, max |x[r] + pos[r % rows]|) for x (R, C) with unit column stride   (gvl_row_absmax_f32)"""
    R, C = x.shape
    assert x.dtype == torch.float32 and x.stride(1) == 1
    am = torch.empty(R, device=x.device, dtype=torch.float32) if want_x else None
    amp = torch.empty(R, device=x.device, dtype=torch.float32) if pos is not None else None
    with torch.cuda.device(x.device):
        rc = _lib.lib().gvl_row_absmax_f32(
            x.data_ptr(), x.stride(0), R, C, pos.data_ptr() if pos is not None else None,
            pos.stride(0) if pos is not None else 0, pos.shape[0] if pos is not None else 0,
            am.data_ptr() if am is not None else None, amp.data_ptr() if amp is not None else None,
            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "row_absmax")
    return am, amp


def tag_amax(t, am):
    """leave the row maxima `am` of tensor `t` on it for the next consumer, valid for THIS version of the data: an
    in-place edit of `t` (a mask, a scale, a user hook) bumps `t._version` and the consumer recomputes (a
    stale, too-small maximum would scale a row's fp16 split out of range, silently)"""
    t._gvl_amax = (am, t._version)
    return t


def amax_of(t, rows):
    """the row maxima left by tag_amax, or None when absent / of another shape / `t` was written since"""
    hit = getattr(t, "_gvl_amax", None)
    if hit is None:
        return None
    am, version = hit
    if am is None or am.numel() != rows or version != t._version:
        return None
    return am


def enc_ref_of(valid_ratios):
    """the encoder's reference points left on `valid_ratios` by prepare_encoder_inputs (one launch produces both), or
    None when absent or `valid_ratios` was written since"""
    hit = getattr(valid_ratios, "_gvl_enc_ref", None)
    if hit is None or hit[1] != valid_ratios._version:
        return None
    return hit[0]


class _Arena:
    """zero-initialised row-maximum vectors for the producers that use atomic max: ONE fill per forward"""

    def __init__(self, n, device):
        self.buf = torch.zeros(n, device=device, dtype=torch.float32)
        self.used = 0

    def take(self, n):
        n16 = (n + 3) // 4 * 4
        assert self.used + n16 <= self.buf.numel()
        out = self.buf[self.used:self.used + n]
        self.used += n16
        return out


def _value_flags():
    """launch flags of the products that write `value` slabs: head m's columns on XCD m (GVL_VALUE_XCD=1; off by default: measured no effect on the sampling launch)"""
    return LIN_XCD_COLUMNS if os.environ.get("GVL_VALUE_XCD", "0") == "1" else 0


def _new(rows, cols, like):
    return torch.empty(rows, cols, device=like.device, dtype=torch.float32)


# ---- eligibility ------------------------------------------------------------------------------------------------------
def _attn_ok(att, host_lengths, S):
    return (att.fused and att.pad_mode in ("zeros", "border") and host_lengths is not None
            and att.d_model // att.n_heads == 64 and att.n_levels * att.n_points == 16 and att.n_points == 4
            and att.d_model % 64 == 0 and (S <= 600 or S - host_lengths[0][0] <= 600)
            # [sampling_offsets | attention_weights] must stay adjacent columns of the concatenated products after
            # Weights' 128-row block padding (encoder: behind value_proj; decoder: the "proj" pair): other head counts
            # take the PyTorch layers
            and (att.n_heads * att.n_levels * att.n_points) % 128 == 0 and att.d_model % 128 == 0)


def _plain(x):
    return x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled() and not torch.is_grad_enabled()


def encoder_eligible(enc, src, temporal_shapes):
    host = getattr(temporal_shapes, "_gvl_host_lengths", None)
    if not (enabled() and _plain(src) and not enc.training and len(enc.layers) > 0):
        return False
    for layer in enc.layers:
        if not (_attn_ok(layer.self_attn, host, src.shape[1]) and layer.linear1.out_features % 64 == 0
                and layer.activation is torch.nn.functional.relu and src.shape[-1] <= 1024):
            return False
    return True


def decoder_eligible(dec, tgt, src, temporal_shapes):
    host = getattr(temporal_shapes, "_gvl_host_lengths", None)
    if not (enabled() and _plain(tgt) and _plain(src) and not dec.training and 0 < len(dec.layers) <= 4):
        return False
    for layer in dec.layers:
        sa = layer.self_attn
        if not (_attn_ok(layer.cross_attn, host, src.shape[1]) and layer.linear1.out_features % 64 == 0
                and layer.activation is torch.nn.functional.relu and tgt.shape[-1] <= 1024
                and sa._qkv_same_embed_dim and sa.in_proj_bias is not None and not sa.batch_first
                and sa.bias_k is None and not sa.add_zero_attn):
            return False
    return True


def encoder_geometry(mask_flatten, lengths, starts, want_ref=True):
    """gvl_encoder_geometry_f32: flattened padding mask (B, S) -> (valid_ratios (B, L), encoder reference points
    (B, S, L, 1) or None)   (deformable_transformer.py:81-83, 209-218)"""
    B, S = mask_flatten.shape
    L = len(lengths)
    m8 = mask_flatten.contiguous().view(torch.uint8)
    vr = torch.empty(B, L, device=m8.device, dtype=torch.float32)
    ref = torch.empty(B, S, L, 1, device=m8.device, dtype=torch.float32) if want_ref else None
    arr = (ctypes.c_int64 * L)
    with torch.cuda.device(m8.device):
        rc = _lib.lib().gvl_encoder_geometry_f32(m8.data_ptr(), B, S, L, arr(*lengths), arr(*starts), vr.data_ptr(),
                                                 ref.data_ptr() if want_ref else None,
                                                 torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "encoder_geometry")
    return vr, ref


# ---- encoder ------------------------------------------------------------------------------------------------------------
def _msda(att, value, proj, ref, shapes2d, lsi, B, Lq, arena):
    am_o = arena.take(B * Lq)
    M = att.n_heads
    # (proj of Lq rows only: the same offsets / logits for every video -- _first_layer_constants)
    o = MSDA.msda1d_fused_forward(value.view(B, -1, M, att.d_model // M), shapes2d, lsi,
                                  proj.view(1 if proj.shape[0] == Lq and B > 1 else B, Lq, -1),
                                  ref.contiguous(), att.n_levels, att.n_points, att.pad_mode, amax_out=am_o)
    return o.view(B * Lq, -1), am_o


def _ffn(layer, x, am_x, norm, arena, pos=None):
    """x -> norm(x + linear2(relu(linear1(x))))   (deformable_transformer.py:189-191 / :257-261)"""
    R, C = x.shape
    h = _new(R, layer.linear1.out_features, x)
    am_h = arena.take(R)
    linear(x, cached(layer, "l1", [(layer.linear1.weight, layer.linear1.bias)]),
           [seg(0, h, am_x, relu=True, amax_out=am_h)])
    y = _new(R, C, x)
    linear(h, cached(layer, "l2", [(layer.linear2.weight, layer.linear2.bias)]), [seg(0, y, am_h, resid=x)])
    return layer_norm(y, norm, pos=pos)


def encoder_forward(enc, src, temporal_shapes, level_start_index, valid_ratios, pos, padding_mask):
    """DeformableTransformerEncoder.forward (deformable_transformer.py:202-226) for inference: 8 launches per layer
    ([value_proj | offsets ; weights] product, sampling, output_proj + residual, LayerNorm, linear1 + ReLU, linear2 +
    residual, LayerNorm) -> memory (B, S, C); its row maxima travel along as ``memory._gvl_amax``."""
    from .ops.modules.ms_deform_attn import temporal_shapes_2d
    B, S, C = src.shape
    R = B * S
    ref = enc_ref_of(valid_ratios)                        # left by prepare_encoder_inputs
    if ref is None or ref.shape[:2] != src.shape[:2]:
        ref = enc.get_reference_points(temporal_shapes, valid_ratios, device=src.device)      # (B, S, L, 1)
    shapes2d = temporal_shapes_2d(temporal_shapes, level_start_index)
    x = src.reshape(R, C).contiguous()
    posf = pos.reshape(R, C).contiguous() if pos is not None else None
    am_x, am_xp = row_absmax(x, posf)
    if posf is None:
        am_xp = am_x
    mask = padding_mask.reshape(R).contiguous().view(torch.uint8) if padding_mask is not None else None
    arena = _Arena(2 * len(enc.layers) * (R + 4), src.device)
    for layer in enc.layers:
        att = layer.self_attn
        n_proj = 2 * att.n_heads * att.n_levels * att.n_points
        w = cached(att, "vp", [(att.value_proj.weight, att.value_proj.bias),
                               (att.sampling_offsets.weight, att.sampling_offsets.bias),
                               (att.attention_weights.weight, att.attention_weights.bias)])
        value, proj = _new(R, C, x), _new(R, n_proj, x)
        linear(x, w, [seg(0, value, am_x, rowmask=mask), seg(w.starts[1], proj, am_xp, addend=posf is not None)], a2=posf,
               flags=_value_flags())
        o, am_o = _msda(att, value, proj, ref, shapes2d, level_start_index, B, S, arena)
        y = _new(R, C, x)
        linear(o, cached(att, "op", [(att.output_proj.weight, att.output_proj.bias)]), [seg(0, y, am_o, resid=x)])
        x1, am1, _ = layer_norm(y, layer.norm1)
        x, am_x, am_xp = _ffn(layer, x1, am1, layer.norm2, arena, pos=posf)
        if posf is None:
            am_xp = am_x
    memory = x.view(B, S, C)
    return tag_amax(memory, am_x)


# ---- decoder ------------------------------------------------------------------------------------------------------------
def mlp_forward(mlp, x, am_x, arena, extra=None):
    """pdvc.py:1166-1178 MLP on rows x (R, C): Linear + ReLU ... Linear -> (R, out_dim).  `extra` = (name, weight, bias): a
    further linear map of x computed by the first launch (the class head shares the box MLP's input) -> (out, extra_out)"""
    R = x.shape[0]
    layers = list(mlp.layers)
    cur, am = x, am_x
    extra_out = None
    for i, lin in enumerate(layers):
        last = i == len(layers) - 1
        pairs = [(lin.weight, lin.bias)]
        with_extra = i == 0 and extra is not None
        if with_extra:
            pairs.append((extra[1], extra[2]))
        w = cached(mlp, f"mlp{i}" + (extra[0] if with_extra else ""), pairs)
        out = _new(R, lin.out_features, x)
        am_o = None if last else arena.take(R)
        segs = [seg(0, out, am, relu=not last, amax_out=am_o, width=lin.out_features)]
        if with_extra:
            extra_out = _new(R, extra[1].shape[0], x)
            segs.append(seg(w.starts[1], extra_out, am, width=extra[1].shape[0]))
        linear(cur, w, segs)
        cur, am = out, am_o
    if extra_out is not None:
        return cur, extra_out
    return cur


def box_refine(delta, ref, valid_ratios, B, Q, want_ref_in=True):
    """gvl_box_refine_f32: delta (R, >= 2), ref (B, Q, RD) -> (sigmoid(delta + inverse_sigmoid(ref)) (B, Q, 2),
    that times the valid ratios (B, Q, L, 2): the reference points of the next layer)"""
    R, L = B * Q, valid_ratios.shape[1]
    ref = ref.contiguous()
    vr = valid_ratios.contiguous()
    new_ref = torch.empty(B, Q, 2, device=delta.device, dtype=torch.float32)
    ref_in = torch.empty(B, Q, L, 2, device=delta.device, dtype=torch.float32) if want_ref_in else None
    with torch.cuda.device(delta.device):
        rc = _lib.lib().gvl_box_refine_f32(delta.data_ptr(), delta.stride(0), ref.data_ptr(), ref.shape[-1], vr.data_ptr(),
                                           B, Q, L, new_ref.data_ptr(), ref_in.data_ptr() if want_ref_in else None,
                                           torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "box_refine")
    return new_ref, ref_in


class _BoxRefineTrain(torch.autograd.Function):
    """sigmoid(delta + inverse_sigmoid(ref)) of the decoder's iterative refinement (deformable_transformer.py:314-324) as ONE
    node on gvl_box_refine_f32 / gvl_box_refine_backward_f32, with the next layer's scaled reference points (:301-304, computed
    from the detached result as there) as a by-product: 2 launches forward + backward where the PyTorch formulation takes ~21
    (four clamps, rsub, div, log, add, cat, sigmoid, stack, mul and their gradients)."""

    @staticmethod
    def forward(ctx, delta, ref, valid_ratios, want_ref_in):
        B, Q = ref.shape[:2]
        d2 = delta.reshape(B * Q, delta.shape[-1])
        if d2.stride(1) != 1:
            d2 = d2.contiguous()
        refc = ref.contiguous()
        new_ref, ref_in = box_refine(d2, refc, valid_ratios, B, Q, want_ref_in)
        ctx.save_for_backward(new_ref, refc)
        ctx.dshape = delta.shape
        if ref_in is None:
            ref_in = new_ref.new_empty(0)
        ctx.mark_non_differentiable(ref_in)
        ctx.set_materialize_grads(False)          # (the non-differentiable by-product would otherwise get a zeros() launch in backward)
        return new_ref, ref_in

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_new, _g_in):
        new_ref, ref = ctx.saved_tensors
        R, RD = new_ref.shape[0] * new_ref.shape[1], ref.shape[-1]
        g = g_new.contiguous()
        gd = torch.empty(R, 2, device=g.device, dtype=torch.float32)
        gr = torch.empty_like(ref) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(g.device):
            rc = _lib.lib().gvl_box_refine_backward_f32(g.data_ptr(), new_ref.data_ptr(), ref.data_ptr(), RD, R, gd.data_ptr(),
                                                        gr.data_ptr() if gr is not None else None,
                                                        torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "box_refine_backward")
        gdelta = gd.view(*ctx.dshape[:-1], 2)
        if ctx.dshape[-1] != 2:                                        # (a wider head output: only its first two columns are read)
            full = gd.new_zeros(ctx.dshape)
            full[..., :2] = gdelta
            gdelta = full
        return gdelta, gr, None, None


def box_refine_train_eligible(delta, ref):
    return (delta.is_cuda and delta.dtype == torch.float32 and ref.dtype == torch.float32 and torch.is_grad_enabled()
            and not torch.is_autocast_enabled() and delta.dim() == 3 and ref.dim() == 3 and ref.shape[-1] in (1, 2)
            and delta.shape[-1] >= 2 and delta.shape[:2] == ref.shape[:2] and os.environ.get("GVL_BOX_REFINE", "") != "torch")


def box_refine_train(delta, ref, valid_ratios, want_ref_in):
    """-> (new_ref (B, Q, 2) attached to delta / ref, the next layer's reference points (B, Q, L, 2) or None)"""
    new_ref, ref_in = _BoxRefineTrain.apply(delta, ref, valid_ratios, want_ref_in)
    return new_ref, (ref_in if want_ref_in else None)


def count_head_eligible(counter, hs):
    return (enabled() and _plain(hs) and hs.dim() == 3 and hs.is_contiguous() and isinstance(counter, torch.nn.Linear)
            and counter.weight.dtype == torch.float32 and hs.shape[-1] % 4 == 0 and hs.shape[-1] <= 2048)


def count_head(counter, hs):
    """predict_event_num (pdvc.py:316-319): counter(max over the queries of hs (B, Q, C)) in one launch"""
    B, Q, C = hs.shape
    out = torch.empty(B, counter.out_features, device=hs.device, dtype=torch.float32)
    with torch.cuda.device(hs.device):
        rc = _lib.lib().gvl_count_head_f32(hs.data_ptr(), B, Q, C, counter.weight.data_ptr(),
                                           counter.bias.data_ptr() if counter.bias is not None else None,
                                           counter.out_features, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "count_head")
    return out


class _CountPoolTrain(torch.autograd.Function):
    """max over the queries of hs (B, Q, C) -> (B, C) (pdvc.py:317, `torch.max(hs_lid, dim=1)`) on gvl_count_pool_f32, and its
    gradient -- the selected row of each column alone -- as ONE dense write (gvl_count_pool_backward_f32) where amax's backward is
    eq, sum, div and mul over the (B, Q, C) tensor"""

    @staticmethod
    def forward(ctx, hs):
        B, Q, C = hs.shape
        pooled = torch.empty(B, C, device=hs.device, dtype=torch.float32)
        arg = torch.empty(B, C, device=hs.device, dtype=torch.int32)
        with torch.cuda.device(hs.device):
            rc = _lib.lib().gvl_count_pool_f32(hs.data_ptr(), B, Q, C, pooled.data_ptr(), arg.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "count_pool")
        ctx.save_for_backward(arg)
        ctx.Q = Q
        return pooled

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        arg, = ctx.saved_tensors
        B, C = arg.shape
        g = g.contiguous()
        dx = torch.empty(B, ctx.Q, C, device=g.device, dtype=torch.float32)
        with torch.cuda.device(g.device):
            rc = _lib.lib().gvl_count_pool_backward_f32(g.data_ptr(), arg.data_ptr(), B, ctx.Q, C, None, None, dx.data_ptr(),
                                                        torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "count_pool_backward")
        return dx


class _ClassCountHeads(torch.autograd.Function):
    """the two heads that read hs (B, Q, C) row by row without a matrix product worth the name -- the single-class head
    (pdvc.py:455: Linear(C, 1)) and the count head's pooling (pdvc.py:317) -- as ONE node: (logits (B, Q, 1), pooled (B, C)).
    Backward: both input gradients leave as one tensor from one launch (gvl_count_pool_backward_f32 with grad_row / w_row) where
    the class head's alone is a rank-1 product written out (9.8 MB at cfg A) and added to the pooling's by autograd."""

    @staticmethod
    def forward(ctx, hs, weight, bias):
        B, Q, C = hs.shape
        logits = torch.addmm(bias, hs.view(B * Q, C), weight.t()).view(B, Q, 1) if bias is not None \
            else torch.mm(hs.view(B * Q, C), weight.t()).view(B, Q, 1)
        pooled = torch.empty(B, C, device=hs.device, dtype=torch.float32)
        arg = torch.empty(B, C, device=hs.device, dtype=torch.int32)
        with torch.cuda.device(hs.device):
            rc = _lib.lib().gvl_count_pool_f32(hs.data_ptr(), B, Q, C, pooled.data_ptr(), arg.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "count_pool")
        ctx.save_for_backward(hs, weight, arg)
        ctx.has_bias = bias is not None
        ctx.set_materialize_grads(False)
        return logits, pooled

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_logits, g_pooled):
        hs, weight, arg = ctx.saved_tensors
        B, Q, C = hs.shape
        if g_logits is None and g_pooled is None:
            return None, None, None
        gl = g_logits.contiguous() if g_logits is not None else None
        gp = g_pooled.contiguous() if g_pooled is not None else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(B, Q, C, device=hs.device, dtype=torch.float32)
            with torch.cuda.device(hs.device):
                rc = _lib.lib().gvl_count_pool_backward_f32(
                    gp.data_ptr() if gp is not None else None, arg.data_ptr(), B, Q, C, gl.data_ptr() if gl is not None else None,
                    weight.data_ptr() if gl is not None else None, dx.data_ptr(), torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "count_pool_backward")
        gw = gb = None
        if gl is not None:
            if ctx.needs_input_grad[1]:
                gw = torch.mm(gl.view(1, B * Q), hs.view(B * Q, C))
            if ctx.has_bias and ctx.needs_input_grad[2]:
                gb = gl.sum().view(1)
        return dx, gw, gb


def class_count_heads_eligible(class_head, hs):
    return (count_pool_train_eligible(hs) and isinstance(class_head, torch.nn.Linear) and class_head.out_features == 1
            and class_head.weight.dtype == torch.float32 and class_head.weight.is_contiguous()
            and class_head.weight.data_ptr() % 16 == 0 and os.environ.get("GVL_CLASS_COUNT_HEADS", "") != "torch")


def class_count_heads(class_head, hs):
    """-> (class logits (B, Q, 1), max over the queries (B, C))"""
    return _ClassCountHeads.apply(hs, class_head.weight, class_head.bias)


class _ExpandParts(torch.autograd.Function):
    """the column blocks of an embedding (Q, parts * C), each expanded over the batch -> parts views (B, Q, C) (batch stride 0, as
    `chunk(...)[h].unsqueeze(0).expand(B, -1, -1)` of deformable_transformer.py:128-135); their gradients are summed over the
    batch into the embedding's in ONE launch (gvl_batch_sum_f32) where expand's backward is a sum(0) per block -- 28 us each at
    (16, 300, 512) -- and a cat"""

    @staticmethod
    def forward(ctx, embed, B, parts):
        Q, PC = embed.shape
        C = PC // parts
        ctx.cfg = (B, parts, Q, C)
        ctx.set_materialize_grads(False)
        # (+ the blocks themselves, un-expanded, for a reader of the embedding's rows: the reference-point head)
        return tuple(embed[:, h * C:(h + 1) * C].unsqueeze(0).expand(B, -1, -1) for h in range(parts)) + tuple(
            embed[:, h * C:(h + 1) * C] for h in range(parts))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        import ctypes
        B, parts, Q, C = ctx.cfg
        rows = grads[parts:]
        grads = [g.contiguous() if g is not None else None for g in grads[:parts]]
        ref = next((g for g in grads if g is not None), None)
        if ref is None:                                                  # (only the un-expanded blocks were used)
            out = next(g for g in rows if g is not None).new_zeros(Q, parts * C)
            for h, g in enumerate(rows):
                if g is not None:
                    out[:, h * C:(h + 1) * C] = g
            return out, None, None
        out = torch.empty(Q, parts * C, device=ref.device, dtype=torch.float32)
        ptrs = (ctypes.c_void_p * parts)(*[g.data_ptr() if g is not None else None for g in grads])
        with torch.cuda.device(ref.device):
            rc = _lib.lib().gvl_batch_sum_f32(ptrs, parts, B, Q, C, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "batch_sum")
        for h, g in enumerate(rows):
            if g is not None:
                out[:, h * C:(h + 1) * C] += g
        return out, None, None


class _EmbedRows(torch.autograd.Function):
    """weight.index_select(0, ids) (the captioner's `self.embed(it)`, LSTM_DSA.py:96, for all teacher-forced steps at once) whose
    backward skips the all-zero gradient rows (gvl_index_add_rows_f32): about half of a padded caption batch's positions are <pad>,
    carry no gradient, and all land on ONE row of the table -- 50 us of serialised atomics in index_add at cfg A."""

    @staticmethod
    def forward(ctx, weight, ids):
        ctx.save_for_backward(ids)
        ctx.shape = weight.shape
        return weight.index_select(0, ids)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        ids, = ctx.saved_tensors
        V, E = ctx.shape
        if g.stride(1) != 1 or g.stride(0) % 4 or g.data_ptr() % 16:
            g = g.contiguous()
        out = torch.zeros(V, E, device=g.device, dtype=torch.float32)
        with torch.cuda.device(g.device):
            rc = _lib.lib().gvl_index_add_rows_f32(g.data_ptr(), g.stride(0), ids.data_ptr(), ids.numel(), E, out.data_ptr(), E, V,
                                                   torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "index_add_rows")
        return out, None


def embed_rows(weight, ids):
    """weight (V, E)[ids (n)] -> (n, E); training on fp32 CUDA tables: through _EmbedRows"""
    if (weight.is_cuda and weight.dtype == torch.float32 and torch.is_grad_enabled() and weight.requires_grad and weight.dim() == 2
            and weight.is_contiguous() and weight.shape[1] % 4 == 0 and ids.dtype == torch.int64 and ids.dim() == 1
            and not torch.is_autocast_enabled() and os.environ.get("GVL_EMBED_ROWS", "") != "torch"):
        return _EmbedRows.apply(weight, ids.contiguous())
    return weight.index_select(0, ids)


class _MaskRows(torch.autograd.Function):
    """value.masked_fill(padding_mask[..., None], 0) (ms_deform_attn.py:100) on the FRESH output of value_proj, in place: only the
    padded rows are written (gvl_mask_rows_f32) where the out-of-place op copies the tensor and passes over it again; backward: the
    gradient with the same rows zeroed and its row maxima in one pass (copy + masked_fill_ + row_absmax before)."""

    @staticmethod
    def forward(ctx, value, mask_u8):
        C = value.shape[-1]
        with torch.cuda.device(value.device):
            rc = _lib.lib().gvl_mask_rows_f32(value.data_ptr(), mask_u8.data_ptr(), value.numel() // C, C,
                                              torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mask_rows")
        ctx.mark_dirty(value)
        ctx.save_for_backward(mask_u8)
        return value

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        mask_u8, = ctx.saved_tensors
        g = g.contiguous()
        C = g.shape[-1]
        R = g.numel() // C
        dx = torch.empty_like(g)
        am = torch.empty(R, device=g.device, dtype=torch.float32)
        with torch.cuda.device(g.device):
            rc = _lib.lib().gvl_mask_rows_backward_f32(g.data_ptr(), mask_u8.data_ptr(), R, C, dx.data_ptr(), am.data_ptr(),
                                                       torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mask_rows_backward")
        return tag_amax(dx, am), None


def mask_rows(value, padding_mask):
    """value (B, S, C) with the rows where padding_mask (B, S) is True set to zero.  Training on fp32 CUDA tensors that own their
    storage (the result of a Linear): in place through _MaskRows; anything else: masked_fill"""
    if (value.is_cuda and value.dtype == torch.float32 and torch.is_grad_enabled() and value.requires_grad and not value.is_leaf
            and value.is_contiguous() and value._base is None and value.dim() == 3 and value.shape[-1] % 4 == 0
            and padding_mask.dtype == torch.bool and padding_mask.shape == value.shape[:2] and value.data_ptr() % 16 == 0
            and not torch.is_autocast_enabled() and os.environ.get("GVL_MASK_ROWS", "") != "torch"):
        return _MaskRows.apply(value, padding_mask.contiguous().view(torch.uint8))
    return value.masked_fill(padding_mask[..., None], float(0))


class _LevelPosEmbed(torch.autograd.Function):
    """lvl_pos_embed_flatten of deformable_transformer.py:96-103 -- cat_l(pos_embed_l^T + level_embed[l]) (B, S, C) -- as one node
    (the position embeddings' gradients -- their learned duration half -- are slices of the incoming one).  The level embedding's is the per-level sum of the (B, S, C)
    gradient over videos and rows in TWO launches (gvl_level_sums_f32 + gvl_batch_sum_f32); the PyTorch formulation reduces each
    level's slice over (0, 1) with a 512-wide output -- 18-31 us apiece at (16, 13..100, 512) -- and stacks the four."""

    @staticmethod
    def forward(ctx, level_embed, *pos_embeds):
        lengths = [int(p.shape[-1]) for p in pos_embeds]
        ctx.lengths, ctx.n_embed = lengths, level_embed.shape[0]
        return torch.cat([p.transpose(1, 2) + level_embed[l].view(1, 1, -1) for l, p in enumerate(pos_embeds)], 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        import ctypes
        g = g.contiguous()
        B, S, C = g.shape
        Lv = len(ctx.lengths)
        starts = [sum(ctx.lengths[:l]) for l in range(Lv)]
        part = torch.empty(B, Lv, C, device=g.device, dtype=torch.float32)
        out = torch.empty(Lv, C, device=g.device, dtype=torch.float32)
        with torch.cuda.device(g.device):
            st = torch.cuda.current_stream().cuda_stream
            rc = _lib.lib().gvl_level_sums_f32(g.data_ptr(), B, S, C, (ctypes.c_int * Lv)(*starts), (ctypes.c_int * Lv)(*ctx.lengths),
                                               Lv, part.data_ptr(), st)
            _lib.check(rc, "level_sums")
            rc = _lib.lib().gvl_batch_sum_f32((ctypes.c_void_p * 1)(part.data_ptr()), 1, B, Lv, C, out.data_ptr(), st)
            _lib.check(rc, "batch_sum")
        if ctx.n_embed > Lv:                                             # (levels of the embedding that this forward did not use)
            out = torch.cat([out, out.new_zeros(ctx.n_embed - Lv, C)])
        grads = []
        for l, (s, n, need) in enumerate(zip(starts, ctx.lengths, ctx.needs_input_grad[1:])):
            gp = g[:, s:s + n].transpose(1, 2) if need else None
            if gp is not None:
                # its sum over the level's rows is at hand: left on the tensor for the position embedding's backward, whose
                # learned half wants exactly that (gvl_amd.base_encoder._PosEmbedSine); versioned like the row maxima
                gp._gvl_time_sum = (part[:, l], gp._version)
            grads.append(gp)
        return (out,) + tuple(grads)


def level_pos_embed_eligible(level_embed, pos_embeds):
    return (level_embed.is_cuda and level_embed.dtype == torch.float32 and torch.is_grad_enabled() and level_embed.requires_grad
            and not torch.is_autocast_enabled() and level_embed.shape[1] % 4 == 0 and 1 <= len(pos_embeds) <= 8
            and level_embed.shape[0] >= len(pos_embeds)
            and all(p.dtype == torch.float32 and p.dim() == 3 for p in pos_embeds)
            and os.environ.get("GVL_LEVEL_POS", "") != "torch")


def level_pos_embed(level_embed, pos_embeds):
    return _LevelPosEmbed.apply(level_embed, *pos_embeds)


def expand_parts_eligible(embed, parts):
    return (embed.is_cuda and embed.dtype == torch.float32 and embed.dim() == 2 and embed.is_contiguous()
            and torch.is_grad_enabled() and embed.requires_grad and not torch.is_autocast_enabled()
            and embed.shape[1] % (4 * parts) == 0 and os.environ.get("GVL_EXPAND_PARTS", "") != "torch")


def expand_parts(embed, B, parts, rows=False):
    """-> the `parts` batch-expanded blocks; rows=True: + the `parts` un-expanded blocks (Q, C) (their gradients join the same
    embedding gradient without a zero-padded cat)"""
    out = _ExpandParts.apply(embed, B, parts)
    return out if rows else out[:parts]


def count_pool_train_eligible(hs):
    return (hs.is_cuda and hs.dtype == torch.float32 and hs.dim() == 3 and hs.is_contiguous() and hs.shape[-1] % 4 == 0
            and torch.is_grad_enabled() and hs.requires_grad and not torch.is_autocast_enabled() and hs.data_ptr() % 16 == 0
            and os.environ.get("GVL_COUNT_POOL", "") != "torch")


def count_pool_train(hs):
    return _CountPoolTrain.apply(hs)


def mha_core(qkv, B, Q, H, key_keep=None, amax_out=None):
    """gvl_mha_core_f32: qkv (B*Q, 3*H*64) rows [q | k | v] -> softmax(q k^T / 8) v per head, (B*Q, H*64); key_keep (B, Q)
    bool, True = the key takes part (the complement of nn.MultiheadAttention's key_padding_mask)"""
    R = B * Q
    assert qkv.shape == (R, 3 * H * 64) and qkv.dtype == torch.float32 and qkv.stride(1) == 1 and Q <= 320
    out = torch.empty(R, H * 64, device=qkv.device, dtype=torch.float32)
    keep = key_keep.contiguous().view(torch.uint8) if key_keep is not None else None
    with torch.cuda.device(qkv.device):
        rc = _lib.lib().gvl_mha_core_f32(qkv.data_ptr(), qkv.stride(0), keep.data_ptr() if keep is not None else None, B, Q, H,
                                         out.data_ptr(), amax_out.data_ptr() if amax_out is not None else None,
                                         torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "mha_core")
    return out


def _self_attention(layer, x, am_x, am_xp, qpos, B, Q, query_mask, arena):
    """nn.MultiheadAttention of the decoder layer (deformable_transformer.py:266-270): q = k = x + query_pos, v = x ->
    (attention output rows (R, C) BEFORE out_proj, their row maxima).  The in-projection is one launch (q, k columns
    multiply x + query_pos, the v columns x); the attention core is gvl_mha_core_f32 (head dimension 64, <= 320 queries)
    or torch's fused SDPA kernel."""
    sa = layer.self_attn
    R, C = x.shape
    H = sa.num_heads
    w = cached(sa, "in", [(sa.in_proj_weight, sa.in_proj_bias)])
    qkv = _new(R, 3 * C, x)
    own = C == 64 * H and B * H * Q * Q < 2 ** 32 and os.environ.get("GVL_MHA", "") != "torch"
    am_qk, am_v = (arena.take(R), arena.take(R)) if own else (None, None)
    linear(x, w, [seg(0, qkv[:, :2 * C], am_xp, addend=qpos is not None, amax_out=am_qk),
                  seg(2 * C, qkv[:, 2 * C:], am_x, amax_out=am_v)], a2=qpos)
    if own:
        # the attention core: the forward kernel of gvl_mha_train.hip at p = 0 (split-fp16 products, scores in registers; it
        # replaced the exact-fp32 MFMA kernel gvl_mha_core_f32 here: 35 against 55 us per layer at B = 16, Q = 300)
        am_a = arena.take(R)
        a = _new(R, C, x)
        keep = query_mask.contiguous().view(torch.uint8) if query_mask is not None else None
        with torch.cuda.device(x.device):
            rc = _lib.lib().gvl_mha_train_forward_f32(qkv.data_ptr(), qkv.stride(0), keep.data_ptr() if keep is not None else None,
                                                      am_qk.data_ptr(), am_v.data_ptr(), B, Q, H, 0.0, 0, None, a.data_ptr(), None,
                                                      am_a.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mha_forward")
        return a, am_a
    t = qkv.view(B, Q, 3, H, C // H).permute(2, 0, 3, 1, 4)                      # (3, B, H, Q, D)
    mask = query_mask[:, None, None, :] if query_mask is not None else None       # True = attend (key_padding_mask = ~)
    o = torch.nn.functional.scaled_dot_product_attention(t[0], t[1], t[2], attn_mask=mask)
    a = o.transpose(1, 2).reshape(R, C)
    return a, row_absmax(a)[0]


def _query_rows(dec, tgt, query_pos, B, Q, C):
    """'queries' input (deformable_transformer.py:128-135): tgt and query_pos are the two halves of the query embedding, expanded
    over the batch -- the decoder's first residual rows, the positional rows and the row maxima of both depend on PARAMETERS
    only.  Kept on the decoder until the embedding changes (its address or its version counter): three launches (two copies of
    9.8 MB / 0.6 MB and the maxima) leave every inference forward.  None when the inputs are not of that form.  Nothing
    downstream writes these rows (every layer output is a new tensor)."""
    if (torch.is_grad_enabled() or query_pos is None or tgt.dim() != 3 or query_pos.dim() != 3 or B < 1
            or tgt.stride(0) != 0 or query_pos.stride(0) != 0 or tgt.dtype != torch.float32 or query_pos.dtype != torch.float32
            or tgt.stride(2) != 1 or query_pos.stride(2) != 1):
        return None
    emb = (tgt.data_ptr(), tgt._version, tgt.stride(1), query_pos.data_ptr(), query_pos._version, query_pos.stride(1), str(tgt.device))
    key = (emb, B, Q, C)
    table = dec.__dict__.get("_gvl_query_rows")
    if not isinstance(table, dict):
        table = dec.__dict__["_gvl_query_rows"] = {}
    hit = table.get(key)
    if hit is None:
        x = tgt.reshape(B * Q, C).contiguous()
        qpos = query_pos[0].contiguous()
        am_x, am_xp = row_absmax(x, qpos)
        if torch.cuda.is_current_stream_capturing():
            dec.__dict__["_gvl_query_rows_key"] = None
            return x, qpos, am_x, am_xp                  # (a graph's private memory is not kept)
        # entries of another embedding value are dead; entries of other batch sizes may be constants of live captured graphs
        for k_ in [k_ for k_ in table if k_[0] != emb]:
            del table[k_]
        hit = table[key] = (x, qpos, am_x, am_xp)
    dec.__dict__["_gvl_query_rows_key"] = key
    return hit


def _first_layer_constants(dec, layer, rows, query_mask, B, Q):
    """'queries' input with no padded query (deformable_transformer.py:128-135, pdvc.py:292): the first decoder layer's self-attention
    block sees the SAME rows for every video -- the batch-expanded query embedding -- and no video: q = k = tgt + query_pos, v = tgt,
    out_proj + residual, norm2, and the offsets / attention-weight projection of (norm2 + query_pos) depend on PARAMETERS only.  In
    inference their results are kept on the decoder until one of those parameters changes (five launches, 0.11 ms of the 7.8 ms
    step): computed once by the ordinary full-batch launches, so the kept rows are bit for bit what every forward would compute.
    -> None (not that form), or (key, kept (x2, am2, am2p, proj) | None on a miss: the caller computes and stores them; never
    during a stream capture -- the tensors would be a graph's private memory)."""
    if rows is None or os.environ.get("GVL_FIRST_LAYER_CACHE", "1") == "0":
        return None
    if query_mask is not None:
        tag = getattr(query_mask, "_gvl_all_true", None)
        if tag is None or tag[1] != query_mask._version or tuple(query_mask.shape) != (B, Q):
            return None
    sa, att = layer.self_attn, layer.cross_attn
    ps = [sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, layer.norm2.weight, layer.norm2.bias,
          att.sampling_offsets.weight, att.sampling_offsets.bias, att.attention_weights.weight, att.attention_weights.bias]
    from . import MultiScaleDeformableAttention as MSDA
    # (params part first: entries of other parameter values are dead -- the graphs that could read them are dropped on a parameter
    #  change, gvl_amd.parallel.GraphedEvalForward -- while entries of other batch sizes may be constants of LIVE captured graphs and
    #  are kept)
    rows_key = dec.__dict__.get("_gvl_query_rows_key")
    if rows_key is None:
        return None
    key = ((rows_key[0], tuple((p_.data_ptr(), p_._version) for p_ in ps if p_ is not None)), B, Q, MSDA.f16_products_now(),
           os.environ.get("GVL_MHA", ""))
    table = dec.__dict__.setdefault("_gvl_first_layer", {})
    return key, table.get(key)


def decoder_forward(dec, tgt, reference_points, src, src_temporal_shapes, src_level_start_index, src_valid_ratios,
                    query_pos, src_padding_mask, query_padding_mask, disable_iterative_refine):
    """DeformableTransformerDecoder.forward (deformable_transformer.py:283-335) for inference.  value_proj(memory) of ALL
    layers is one launch up front; per layer: in-projection, attention core, out_proj + residual, LayerNorm, offset /
    weight projection of tgt + query_pos, sampling, output_proj + residual, LayerNorm, FFN (3 launches), box MLP."""
    from .deformable_transformer import inverse_sigmoid
    from .ops.modules.ms_deform_attn import temporal_shapes_2d
    B, Q, C = tgt.shape
    S = src.shape[1]
    R, Rs = B * Q, B * S
    shapes2d = temporal_shapes_2d(src_temporal_shapes, src_level_start_index)
    mem = src.reshape(Rs, C)
    if not mem.is_contiguous():
        mem = mem.contiguous()
    am_mem = amax_of(src, Rs)
    if am_mem is None:
        am_mem, _ = row_absmax(mem)
    mask = src_padding_mask.reshape(Rs).contiguous().view(torch.uint8) if src_padding_mask is not None else None
    nl = len(dec.layers)
    arena = _Arena((7 * nl + 2) * (R + 4), tgt.device)          # (+ the row maxima of the q | k and v columns per layer)
    # the first layer's scaled reference points (:302-306) BEFORE the value product: with the first layer's self-attention block kept
    # across forwards (_first_layer_constants) the sampling launch would otherwise directly follow this few-workgroup launch, and a
    # launch behind one that left most of the chip idle runs 0.5-1 us longer (8.3 against 9.2 us inside the replayed graph, same box,
    # GVL_REF_FIRST=0; tools/fwd_layer_stamps.py shows the two layers' launches)
    first_ref_in = None
    if os.environ.get("GVL_REF_FIRST", "1") != "0":
        if reference_points.shape[-1] == 2:
            first_ref_in = reference_points[:, :, None] * torch.stack([src_valid_ratios] * 2, -1)[:, None]
        else:
            assert reference_points.shape[-1] == 1
            first_ref_in = reference_points[:, :, None] * src_valid_ratios[:, None, :, None]
    # value_proj(memory) of every layer: one product against the concatenated weights
    wv = cached(dec, "values", [(l_.cross_attn.value_proj.weight, l_.cross_attn.value_proj.bias) for l_ in dec.layers])
    values = [_new(Rs, C, mem) for _ in dec.layers]
    linear(mem, wv, [seg(wv.starts[i], values[i], am_mem, rowmask=mask) for i in range(nl)], flags=_value_flags())
    rows = _query_rows(dec, tgt, query_pos, B, Q, C)
    if rows is not None:
        x, qpos, am_x, am_xp = rows
    else:
        x = tgt.reshape(R, C).contiguous()
        if query_pos is None:
            qpos = None
        elif query_pos.stride(0) == 0:                # 'queries' input: the same embedding for every video (:130-133)
            qpos = query_pos[0].contiguous()
        else:
            qpos = query_pos.reshape(R, C).contiguous()
        am_x, am_xp = row_absmax(x, qpos)
        if qpos is None:
            am_xp = am_x
    hs, refs, deltas, coords, clss = [], [], [], [], []
    next_ref_in = None
    for lid, layer in enumerate(dec.layers):
        if next_ref_in is not None:                                               # left by the previous layer's refinement
            ref_in = next_ref_in
        elif lid == 0 and first_ref_in is not None:
            ref_in = first_ref_in
        elif reference_points.shape[-1] == 2:                                     # :302-304
            ref_in = reference_points[:, :, None] * torch.stack([src_valid_ratios] * 2, -1)[:, None]
        else:
            assert reference_points.shape[-1] == 1
            ref_in = reference_points[:, :, None] * src_valid_ratios[:, None, :, None]
        att = layer.cross_attn
        first = _first_layer_constants(dec, layer, rows, query_padding_mask, B, Q) if lid == 0 else None
        if first is not None and first[1] is not None:
            x2, am2, am2p, proj = first[1]
        else:
            # -- self attention over the queries
            a, am_a = _self_attention(layer, x, am_x, am_xp, qpos, B, Q, query_padding_mask, arena)
            sa = layer.self_attn
            y = _new(R, C, x)
            linear(a, cached(sa, "out", [(sa.out_proj.weight, sa.out_proj.bias)]), [seg(0, y, am_a, resid=x)])
            x2, am2, am2p = layer_norm(y, layer.norm2, pos=qpos)
            if qpos is None:
                am2p = am2
            # -- deformable cross attention into the memory
            proj = _new(R, 2 * att.n_heads * att.n_levels * att.n_points, x)
            linear(x2, cached(att, "proj", [(att.sampling_offsets.weight, att.sampling_offsets.bias),
                                            (att.attention_weights.weight, att.attention_weights.bias)]),
                   [seg(0, proj, am2p, addend=qpos is not None)], a2=qpos)
            if first is not None and not torch.cuda.is_current_stream_capturing():
                table = dec.__dict__["_gvl_first_layer"]
                for k_ in [k_ for k_ in table if k_[0] != first[0][0]]:
                    del table[k_]
                # (the projection's rows are the same for every video -- same inputs, every output element its own dot product: the
                #  first video's Q rows are kept and every video reads them, 0.3 instead of 4.9 MB per launch at cfg A)
                keep_q = os.environ.get("GVL_FIRST_LAYER_SHARED_PROJ", "1") != "0" and B > 1
                table[first[0]] = (x2, am2, am2p, proj[:Q].clone() if keep_q else proj)
        o, am_o = _msda(att, values[lid], proj, ref_in, shapes2d, src_level_start_index, B, Q, arena)
        y = _new(R, C, x)
        linear(o, cached(att, "op", [(att.output_proj.weight, att.output_proj.bias)]), [seg(0, y, am_o, resid=x2)])
        x3, am3, _ = layer_norm(y, layer.norm1)
        # -- FFN
        x, am_x, am_xp = _ffn(layer, x3, am3, layer.norm3, arena, pos=qpos)
        if qpos is None:
            am_xp = am_x
        out = x.view(B, Q, C)
        if not disable_iterative_refine and dec.bbox_head is not None:           # :314-324
            cls_heads = dec.__dict__.get("_gvl_class_head")
            ch = cls_heads[lid] if cls_heads is not None and isinstance(cls_heads[lid], torch.nn.Linear) else None
            if ch is not None:                                                    # pdvc.py:452: same input rows
                delta, cls = mlp_forward(dec.bbox_head[lid], x, am_x, arena, extra=("cls", ch.weight, ch.bias))
                clss.append(cls.view(B, Q, -1))
            else:
                delta = mlp_forward(dec.bbox_head[lid], x, am_x, arena)
            deltas.append(delta.view(B, Q, -1))
            # sigmoid(delta + inverse_sigmoid(ref)) and the next layer's scaled reference points in one launch
            new_ref, next_ref_in = box_refine(delta, reference_points, src_valid_ratios, B, Q, lid + 1 < nl)
            coords.append(new_ref)
            reference_points = new_ref
        if dec.return_intermediate:
            hs.append(out)
            refs.append(reference_points)
    # the heads of pdvc.py:452-474 apply the same box MLP / refinement arithmetic to the same rows: handed over (taken once)
    ok = len(deltas) == nl
    dec.__dict__["_gvl_deltas"] = deltas if ok else None
    dec.__dict__["_gvl_coords"] = coords if ok else None
    dec.__dict__["_gvl_cls"] = clss if ok and len(clss) == nl else None
    if dec.return_intermediate:
        hs_t = torch.stack(hs)
        tag_amax(hs_t, am_x)
        return hs_t, torch.stack(refs)
    return out, reference_points
