"""Drop-in for the reference's native extension module ``MultiScaleDeformableAttention``.

The reference builds that module from pdvc/ops/src (setup.py:51-58) and calls
``MSDA.ms_deform_attn_forward / ms_deform_attn_backward`` from ms_deform_attn_func.py:25-41; the two functions
here keep those names, argument order and error behaviour (vision.cpp:13-16, ms_deform_attn.h:20-61,
ms_deform_attn_cuda.cu:20-153) and forward to the C ABI of libgvl_msda.so (include/gvl_msda.h).
Extra keyword ``pad_mode`` ("zeros" = reference CUDA op, "border" = reference CPU fallback semantics).
"""
import ctypes

import numpy as np
import torch

from . import _lib

PAD_MODES = {"zeros": 0, "border": 1, 0: 0, 1: 1}
_SUFFIX = {torch.float32: "f32", torch.float64: "f64", torch.bfloat16: "bf16"}
_BF16_WIDEN = (b"bf16 storage needs", b"does not fit LDS")     # bf16 entry points serve the temporal D=64 kernels only


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)          # AT_ASSERTM surfaces as RuntimeError in Python (cu:28-38)


def host_shapes(spatial_shapes, level_start_index):
    """HOST int64 copies of (shapes (L,2), lsi (L)) -- cached on the tensor objects so that the device->host
    read happens once per tensor, not once per call.  gvl_amd's own modules attach the cache at construction
    time and never synchronise."""
    cached = getattr(spatial_shapes, "_gvl_host", None)
    if cached is None:
        sh = np.ascontiguousarray(spatial_shapes.detach().cpu().numpy().astype(np.int64))
        ls = np.ascontiguousarray(level_start_index.detach().cpu().numpy().astype(np.int64))
        cached = (sh, ls)
        try:
            spatial_shapes._gvl_host = cached
        except Exception:  # pragma: no cover
            pass
    return cached


def attach_host_shapes(spatial_shapes, level_start_index, shapes_list, lsi_list):
    """Record host copies without touching the device (used by gvl_amd modules that build the tensors)."""
    spatial_shapes._gvl_host = (np.ascontiguousarray(np.asarray(shapes_list, dtype=np.int64).reshape(-1, 2)),
                                np.ascontiguousarray(np.asarray(lsi_list, dtype=np.int64)))
    return spatial_shapes


def _common_checks(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    for name, t_ in (("value", value), ("spatial_shapes", spatial_shapes),
                     ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                     ("attn_weight", attn_weight)):
        if t_ is None:
            continue
        _require(t_.is_contiguous(), f"{name} tensor has to be contiguous")
        if not t_.is_cuda:
            raise RuntimeError("Not implemented on the CPU")                        # ms_deform_attn.h:38
    _require(value.dtype in _SUFFIX, f"ms_deform_attn: unsupported dtype {value.dtype} (fp32 / fp64 / bf16 only)")
    _require(spatial_shapes.dtype == torch.int64 and level_start_index.dtype == torch.int64,
             "spatial_shapes / level_start_index must be int64")
    B, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Q, P = sampling_loc.shape[1], sampling_loc.shape[4]
    _require(tuple(sampling_loc.shape) == (B, Q, M, L, P, 2), "sampling_loc has wrong shape")
    if attn_weight is not None:
        _require(tuple(attn_weight.shape) == (B, Q, M, L, P), "attn_weight has wrong shape")
        _require(attn_weight.dtype == _arith_dtype(value), "dtype mismatch")
    _require(sampling_loc.dtype == _arith_dtype(value), "dtype mismatch")
    step = min(B, int(im2col_step)) if B > 0 else 1
    _require(step > 0 and B % step == 0, f"batch({B}) must divide im2col_step({step})")   # cu:50-52
    return B, S, M, D, L, Q, P


def _hp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _arith_dtype(value):
    """bf16 is a STORAGE type here (include/gvl_msda.h "Element types"): locations / weights stay fp32."""
    return torch.float32 if value.dtype == torch.bfloat16 else value.dtype


def _bf16_unserved(rc):
    return rc == -1 and any(m in (_lib.lib().gvl_last_error() or b"") for m in _BF16_WIDEN)


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step,
                           pad_mode="zeros"):
    B, S, M, D, L, Q, P = _common_checks(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                         im2col_step)
    sh, ls = host_shapes(spatial_shapes, level_start_index)
    out = value.new_empty((B, Q, M * D))
    fn = getattr(_lib.lib(), "gvl_msda_forward_" + _SUFFIX[value.dtype])
    with torch.cuda.device(value.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = fn(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                attn_weight.data_ptr(), B, S, M, D, L, Q, P, PAD_MODES[pad_mode], _hp(sh), _hp(ls), out.data_ptr(),
                stream)
    if value.dtype == torch.bfloat16 and _bf16_unserved(rc):     # other shapes: widen, same fp32 arithmetic, round once
        return ms_deform_attn_forward(value.float(), spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                      im2col_step, pad_mode).to(torch.bfloat16)
    _lib.check(rc, "ms_deform_attn_forward")
    return out


def ms_deform_attn_sample(value, spatial_shapes, level_start_index, sampling_loc, pad_mode="border"):
    """ms_deform_attn_core_pytorch(..., return_value=True) (func.py:67-68): (B*M, D, Q, L, P)."""
    B, S, M, D, L, Q, P = _common_checks(value, spatial_shapes, level_start_index, sampling_loc, None, 1 << 30)
    if value.dtype == torch.bfloat16:
        return ms_deform_attn_sample(value.float(), spatial_shapes, level_start_index, sampling_loc,
                                     pad_mode).to(torch.bfloat16)
    out = value.new_empty((B * M, D, Q, L, P))
    fn = getattr(_lib.lib(), "gvl_msda_sample_" + _SUFFIX[value.dtype])
    with torch.cuda.device(value.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = fn(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                B, S, M, D, L, Q, P, PAD_MODES[pad_mode], out.data_ptr(), stream)
    _lib.check(rc, "ms_deform_attn_sample")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step, pad_mode="zeros"):
    B, S, M, D, L, Q, P = _common_checks(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                         im2col_step)
    _require(grad_output.is_contiguous(), "grad_output tensor has to be contiguous")   # cu:98
    _require(grad_output.is_cuda, "grad_output must be a CUDA tensor")
    _require(grad_output.dtype == value.dtype, "grad_output dtype mismatch")
    sh, ls = host_shapes(spatial_shapes, level_start_index)
    grad_value = torch.empty_like(value)
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    lib = _lib.lib()
    nbytes = lib.gvl_msda_backward_workspace_bytes(B, S, M, D, L, Q, P, value.element_size(), _hp(sh))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=value.device) if nbytes else None
    fn = getattr(lib, "gvl_msda_backward_" + _SUFFIX[value.dtype])
    with torch.cuda.device(value.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = fn(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                attn_weight.data_ptr(), grad_output.data_ptr(), B, S, M, D, L, Q, P, PAD_MODES[pad_mode], _hp(sh),
                _hp(ls), grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(),
                ws.data_ptr() if ws is not None else None, nbytes, stream)
    if value.dtype == torch.bfloat16 and _bf16_unserved(rc):
        gv, gl, ga = ms_deform_attn_backward(value.float(), spatial_shapes, level_start_index, sampling_loc,
                                             attn_weight, grad_output.float(), im2col_step, pad_mode)
        return gv.to(torch.bfloat16), gl, ga
    _lib.check(rc, "ms_deform_attn_backward")
    return grad_value, grad_loc, grad_attn


def msda1d_fused_forward(value, spatial_shapes, level_start_index, proj, ref, n_levels, n_points, pad_mode="zeros",
                         amax_out=None):
    """include/gvl_msda.h: gvl_msda1d_fused_forward_{f32,bf16}.  value (B,S,M,64) | proj (B,Q,2*M*L*P) | ref (B,Q,L,1|2);
    value / proj fp32 or both bf16, ref always fp32.  amax_out (B*Q) fp32, zero-initialised: additionally receives
    max |out| of every output row (gvl_msda1d_fused_forward_amax_f32; fp32 only).  proj (1, Q, 2*M*L*P) with amax_out: the same
    rows for every video (gvl_msda1d_fused_forward_shared_amax_f32)."""
    _require(value.dtype in (torch.float32, torch.bfloat16), "msda1d_fused: value must be fp32 or bf16")
    for name, t_, dt_ in (("value", value, value.dtype), ("proj", proj, value.dtype), ("ref", ref, torch.float32)):
        _require(t_.is_cuda and t_.is_contiguous() and t_.dtype == dt_,
                 f"msda1d_fused: {name} must be a contiguous {dt_} CUDA tensor")
    B, S, M, D = value.shape
    Q, RD = ref.shape[1], ref.shape[-1]
    shared = B > 1 and proj.shape[0] == 1 and amax_out is not None
    _require(tuple(proj.shape) == (1 if shared else B, Q, 2 * M * n_levels * n_points) and tuple(ref.shape) == (B, Q, n_levels, RD),
             "msda1d_fused: proj / ref have wrong shapes")
    sh, ls = host_shapes(spatial_shapes, level_start_index)
    out = value.new_empty((B, Q, M * D))
    if amax_out is not None:
        _require(value.dtype == torch.float32 and amax_out.is_cuda and amax_out.dtype == torch.float32
                 and amax_out.is_contiguous() and amax_out.numel() == B * Q, "msda1d_fused: amax_out must be (B*Q) fp32")
        with torch.cuda.device(value.device):
            entry = _lib.lib().gvl_msda1d_fused_forward_shared_amax_f32 if shared else _lib.lib().gvl_msda1d_fused_forward_amax_f32
            rc = entry(
                value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), proj.data_ptr(),
                ref.data_ptr(), B, S, M, D, n_levels, Q, n_points, RD, PAD_MODES[pad_mode], _hp(sh), _hp(ls),
                out.data_ptr(), amax_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "msda1d_fused_forward_amax")
        return out
    with torch.cuda.device(value.device):
        rc = getattr(_lib.lib(), "gvl_msda1d_fused_forward_" + _SUFFIX[value.dtype])(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), proj.data_ptr(), ref.data_ptr(),
            B, S, M, D, n_levels, Q, n_points, RD, PAD_MODES[pad_mode], _hp(sh), _hp(ls), out.data_ptr(),
            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "msda1d_fused_forward")
    return out


def msda1d_fused_backward(value, spatial_shapes, level_start_index, proj, ref, grad_output, n_levels, n_points,
                          pad_mode="zeros", need_ref_grad=False):
    """-> (grad_value, grad_proj, grad_ref or None)"""
    B, S, M, D = value.shape
    Q, RD = ref.shape[1], ref.shape[-1]
    _require(grad_output.is_contiguous() and grad_output.is_cuda and grad_output.dtype == value.dtype,
             "grad_output must be a contiguous CUDA tensor of value's dtype")
    sh, ls = host_shapes(spatial_shapes, level_start_index)
    grad_value = torch.empty_like(value)
    grad_proj = torch.empty_like(proj)
    grad_ref_part = ref.new_empty((B, Q, M, n_levels, RD)) if need_ref_grad else None
    lib = _lib.lib()
    nbytes = lib.gvl_msda_backward_workspace_bytes(B, S, M, D, n_levels, Q, n_points, value.element_size(), _hp(sh))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=value.device) if nbytes else None
    with torch.cuda.device(value.device):
        rc = getattr(lib, "gvl_msda1d_fused_backward_" + _SUFFIX[value.dtype])(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), proj.data_ptr(), ref.data_ptr(),
            grad_output.data_ptr(), B, S, M, D, n_levels, Q, n_points, RD, PAD_MODES[pad_mode], _hp(sh), _hp(ls),
            grad_value.data_ptr(), grad_proj.data_ptr(), grad_ref_part.data_ptr() if need_ref_grad else None,
            ws.data_ptr() if ws is not None else None, nbytes, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "msda1d_fused_backward")
    return grad_value, grad_proj, (grad_ref_part.sum(2) if need_ref_grad else None)


def ms_deform_attn_sample_backward(value, spatial_shapes, level_start_index, sampling_loc, grad_sample,
                                   pad_mode="border"):
    """autograd of ms_deform_attn_sample: grad_sample (B*M, D, Q, L, P) -> (grad_value, grad_loc)."""
    B, S, M, D, L, Q, P = _common_checks(value, spatial_shapes, level_start_index, sampling_loc, None, 1 << 30)
    _require(grad_sample.is_contiguous() and tuple(grad_sample.shape) == (B * M, D, Q, L, P),
             "grad_sample must be contiguous (B*M, D, Q, L, P)")
    grad_value = torch.empty_like(value)
    grad_loc = torch.empty_like(sampling_loc)
    fn = getattr(_lib.lib(), "gvl_msda_sample_backward_" + _SUFFIX[value.dtype])
    with torch.cuda.device(value.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = fn(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                grad_sample.data_ptr(), B, S, M, D, L, Q, P, PAD_MODES[pad_mode], grad_value.data_ptr(),
                grad_loc.data_ptr(), stream)
    _lib.check(rc, "ms_deform_attn_sample_backward")
    return grad_value, grad_loc


def cap_attend_pre_applicable(S, n_levels, n_points, host_starts):
    """does the precomputed-offsets form of the token step's attention exist for this pyramid (gvl_cap_attend_pre_applicable)"""
    if host_starts is None or len(host_starts) != n_levels:
        return False
    arr = (ctypes.c_int64 * n_levels)(*[int(v) for v in host_starts])
    return bool(_lib.lib().gvl_cap_attend_pre_applicable(int(S), int(n_levels), int(n_points), ctypes.cast(arr, ctypes.c_void_p)))


def cap_attend_pre(slab, spatial_shapes, level_start_index, ref_in, off_hs, off_pre, att_h, alpha_w, alpha_b, n_levels, n_points,
                   host_starts):
    """cap_attend(planes=True) with the hidden-state part of the offsets precomputed: off_pre (B*Q, >= L*P) fp32, unit column
    stride (gvl_cap_attend_pre_f32; callers check cap_attend_pre_applicable first)"""
    for name, t_ in (("ref_in", ref_in), ("off_hs", off_hs), ("alpha_w", alpha_w)):
        _require(t_.is_cuda and t_.is_contiguous() and t_.dtype == torch.float32,
                 f"cap_attend_pre: {name} must be a contiguous fp32 CUDA tensor")
    _require(slab.is_cuda and slab.is_contiguous() and slab.dtype == torch.float32, "cap_attend_pre: slab must be contiguous fp32")
    for name, t_ in (("att_h", att_h), ("off_pre", off_pre)):
        _require(t_.is_cuda and t_.dtype == torch.float32 and t_.dim() == 2 and t_.stride(1) == 1,
                 f"cap_attend_pre: {name} must be an fp32 CUDA matrix with unit column stride")
    B, S, C2 = slab.shape
    C, Q, RD = C2 // 2, ref_in.shape[1], ref_in.shape[-1]
    _require(off_pre.shape[0] == B * Q and off_pre.shape[1] >= n_levels * n_points, "cap_attend_pre: off_pre must be (B*Q, >= L*P)")
    out = SplitPlanes(B * Q, C, slab.device)
    hs_arr = (ctypes.c_int64 * n_levels)(*[int(v) for v in host_starts])
    with torch.cuda.device(slab.device):
        rc = _lib.lib().gvl_cap_attend_pre_f32(
            slab.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), ref_in.data_ptr(), off_hs.data_ptr(),
            off_pre.data_ptr(), off_pre.stride(0), att_h.data_ptr(), alpha_w.data_ptr(), float(alpha_b), B, S, C, n_levels, Q,
            n_points, RD, att_h.stride(0), ctypes.cast(hs_arr, ctypes.c_void_p), out.hi.data_ptr(), out.lo.data_ptr(),
            out.scale.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "cap_attend_pre")
    return out


def cap_attend(slab, spatial_shapes, level_start_index, ref_in, off_hs, h, w_off_h, att_h, alpha_w, alpha_b,
               n_levels, n_points, debug=False, planes=False, host_starts=None):
    """Fused deformable soft attention of one captioner token step (include/gvl_msda.h: gvl_cap_attend_f32 / _bf16).
    slab (B,S,2C) | ref_in (B,Q,L,1|2) | off_hs (B,Q,L*P) | h, att_h (B*Q,C) | w_off_h (L*P,C) | alpha_w (C,)
    slab and att_h: both fp32 or both bf16 (GEMM outputs under autocast); everything else fp32.
    planes=True (fp32 only): the result as the SplitPlanes operand of gemm_f16x3 (gvl_cap_attend_split_f32);
    host_starts: the level starts as host integers when the caller has them (gvl_cap_attend_split_levels_f32: the coarse
    levels' rows then stay in LDS)."""
    st = slab.dtype
    _require(st in (torch.float32, torch.bfloat16), "cap_attend: slab must be fp32 or bf16")
    for name, t_ in (("ref_in", ref_in), ("off_hs", off_hs), ("h", h), ("w_off_h", w_off_h), ("alpha_w", alpha_w)):
        _require(t_.is_cuda and t_.is_contiguous() and t_.dtype == torch.float32,
                 f"cap_attend: {name} must be a contiguous fp32 CUDA tensor")
    _require(slab.is_cuda and slab.is_contiguous(), "cap_attend: slab must be a contiguous CUDA tensor")
    _require(att_h.is_cuda and att_h.dtype == st and att_h.dim() == 2 and att_h.stride(1) == 1,
             "cap_attend: att_h must be a CUDA matrix of the slab's dtype with unit column stride")
    B, S, C2 = slab.shape
    C = C2 // 2
    Q = ref_in.shape[1]
    RD = ref_in.shape[-1]
    if planes:
        _require(st == torch.float32 and not debug, "cap_attend: planes=True needs fp32 operands (and no debug outputs)")
        out = SplitPlanes(B * Q, C, slab.device)
        hs_arr = None
        if host_starts is not None and len(host_starts) == n_levels:
            hs_arr = (ctypes.c_int64 * n_levels)(*[int(v) for v in host_starts])
        with torch.cuda.device(slab.device):
            rc = _lib.lib().gvl_cap_attend_split_levels_f32(
                slab.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), ref_in.data_ptr(),
                off_hs.data_ptr(), h.data_ptr(), w_off_h.data_ptr(), att_h.data_ptr(), alpha_w.data_ptr(),
                float(alpha_b), B, S, C, n_levels, Q, n_points, RD, att_h.stride(0),
                ctypes.cast(hs_arr, ctypes.c_void_p) if hs_arr is not None else None, out.hi.data_ptr(),
                out.lo.data_ptr(), out.scale.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "cap_attend_split")
        return out
    att_res = torch.empty((B * Q, C), device=slab.device, dtype=st)      # bf16 kernel: bf16 (A operand of a bf16 GEMM)
    dbg_a = torch.empty((B * Q, n_levels * n_points), device=slab.device) if debug else None
    dbg_l = torch.empty((B * Q, n_levels * n_points), device=slab.device) if debug else None
    with torch.cuda.device(slab.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = getattr(_lib.lib(), "gvl_cap_attend_" + _SUFFIX[st])(
            slab.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), ref_in.data_ptr(),
            off_hs.data_ptr(), h.data_ptr(), w_off_h.data_ptr(), att_h.data_ptr(), alpha_w.data_ptr(),
            float(alpha_b), B, S, C, n_levels, Q, n_points, RD, att_h.stride(0), att_res.data_ptr(),
            dbg_a.data_ptr() if debug else None, dbg_l.data_ptr() if debug else None, stream)
    _lib.check(rc, "cap_attend")
    return (att_res, dbg_a, dbg_l) if debug else att_res


def _f32_rows(name, t_, cols):
    _require(t_.is_cuda and t_.dtype == torch.float32 and t_.dim() == 2 and t_.stride(1) == 1 and t_.shape[1] == cols,
             f"{name} must be an fp32 CUDA matrix with {cols} unit-stride columns")


def _cap_rows(ref_in, row_video, B):
    """(Q argument, number of rows, row_video pointer) of the two row layouts of gvl_cap_attend_train_*"""
    if row_video is None:
        _require(ref_in.dim() == 4 and ref_in.shape[0] == B, "cap_attend_train: ref_in must be (B, Q, L, RD)")
        return ref_in.shape[1], B * ref_in.shape[1], None
    _require(ref_in.dim() == 3 and row_video.is_cuda and row_video.dtype == torch.int64 and row_video.is_contiguous()
             and row_video.numel() == ref_in.shape[0],
             "cap_attend_train: compact form needs ref_in (n, L, RD) and row_video (n,) int64 on the device")
    return ref_in.shape[0], ref_in.shape[0], row_video.data_ptr()


def cap_attend_train_forward(slab, spatial_shapes, level_start_index, ref_in, off_hs, off_h, att_h, alpha_w, alpha_b,
                             n_levels, n_points, att_res=None, alpha_out=None, row_video=None):
    """include/gvl_msda.h: gvl_cap_attend_train_forward_f32 -> (att_res (n,C), alpha (n,16)); off_h / att_h may be
    column blocks of one GEMM output (row strides are passed on).  row_video: the compact row form (ref_in (n,L,RD))."""
    B, S, C2 = slab.shape
    C, RD, K = C2 // 2, ref_in.shape[-1], n_levels * n_points
    Q, n, rv = _cap_rows(ref_in, row_video, B)
    for name, t_ in (("slab", slab), ("ref_in", ref_in), ("off_hs", off_hs), ("alpha_w", alpha_w), ("alpha_b", alpha_b)):
        _require(t_.is_cuda and t_.is_contiguous() and t_.dtype == torch.float32,
                 f"cap_attend_train: {name} must be a contiguous fp32 CUDA tensor")
    _f32_rows("cap_attend_train: off_h", off_h, K)
    _f32_rows("cap_attend_train: att_h", att_h, C)
    if att_res is None:
        att_res = torch.empty((n, C), device=slab.device, dtype=torch.float32)
    if alpha_out is None:
        alpha_out = torch.empty((n, 16), device=slab.device, dtype=torch.float32)
    with torch.cuda.device(slab.device):
        rc = _lib.lib().gvl_cap_attend_train_forward_f32(
            slab.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), ref_in.data_ptr(),
            off_hs.data_ptr(), off_h.data_ptr(), off_h.stride(0), att_h.data_ptr(), att_h.stride(0),
            alpha_w.data_ptr(), alpha_b.data_ptr(), B, S, C, n_levels, Q, n_points, RD, rv, att_res.data_ptr(),
            alpha_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "cap_attend_train_forward")
    return att_res, alpha_out


def cap_attend_train_backward(slab, spatial_shapes, level_start_index, ref_in, off_hs, off_h, att_h, alpha_w, alpha,
                              grad_att_res, n_levels, n_points, grad_slab, grad_att_h, grad_off, grad_ref,
                              grad_alpha_w, grad_alpha_b, row_video=None):
    """include/gvl_msda.h: gvl_cap_attend_train_backward_f32.  grad_att_h / grad_off are overwritten (may be column
    blocks of one matrix); grad_slab / grad_ref / grad_alpha_w / grad_alpha_b are accumulated into."""
    B, S, C2 = slab.shape
    C, RD, K = C2 // 2, ref_in.shape[-1], n_levels * n_points
    Q, _, rv = _cap_rows(ref_in, row_video, B)
    _f32_rows("cap_attend_train: grad_att_res", grad_att_res, C)
    _f32_rows("cap_attend_train: grad_att_h", grad_att_h, C)
    _f32_rows("cap_attend_train: grad_off", grad_off, 16)
    for name, t_, shape in (("grad_slab", grad_slab, slab.shape), ("grad_ref", grad_ref, ref_in.shape),
                            ("grad_alpha_w", grad_alpha_w, (C,)), ("grad_alpha_b", grad_alpha_b, (1,))):
        _require(t_.is_cuda and t_.is_contiguous() and t_.dtype == torch.float32 and tuple(t_.shape) == tuple(shape),
                 f"cap_attend_train: {name} must be a contiguous fp32 CUDA tensor of shape {tuple(shape)}")
    with torch.cuda.device(slab.device):
        rc = _lib.lib().gvl_cap_attend_train_backward_f32(
            slab.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), ref_in.data_ptr(),
            off_hs.data_ptr(), off_h.data_ptr(), off_h.stride(0), att_h.data_ptr(), att_h.stride(0),
            alpha_w.data_ptr(), alpha.data_ptr(), grad_att_res.data_ptr(), grad_att_res.stride(0), B, S, C, n_levels,
            Q, n_points, RD, rv, grad_slab.data_ptr(), grad_att_h.data_ptr(), grad_att_h.stride(0), grad_off.data_ptr(),
            grad_off.stride(0), grad_ref.data_ptr(), grad_alpha_w.data_ptr(), grad_alpha_b.data_ptr(),
            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "cap_attend_train_backward")


def proj_eligible(x, weight, bias):
    """domain of gvl_proj_f32: fp32, contiguous, K in {256, 512, 1024}, N a multiple of 64"""
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.is_contiguous()
            and weight.is_contiguous() and x.shape[-1] == weight.shape[1] and weight.shape[1] in (256, 512, 1024)
            and weight.shape[0] % 64 == 0 and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous()))
            and x.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0)


def proj_linear(x, weight, bias=None):
    """x (..., K) @ weight (N, K)^T + bias -> (..., N) through the hand-written fp32 MFMA kernel (include/gvl_msda.h:
    gvl_proj_f32): the offset / attention-logit projection of MSDeformAttn (ms_deform_attn.py:99-100)."""
    _require(proj_eligible(x, weight, bias), "proj_linear: needs contiguous fp32 CUDA operands with K, N multiples of 64")
    K, N = weight.shape[1], weight.shape[0]
    R = x.numel() // K
    out = torch.empty(x.shape[:-1] + (N,), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        rc = _lib.lib().gvl_proj_f32(x.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None, R, K,
                                     N, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "proj")
    return out


class SplitPlanes:
    """an fp32 matrix as two fp16 planes and a row scale: x[r, k] = scale[r] (hi[r, k] + 2^-11 lo[r, k])
    (include/gvl_msda.h: gvl_split_rows_f16).  The planes are stored K-STAGE-MAJOR, (K / 32, rows, 32): the 64 bytes one K
    stage of the GEMM kernels takes from a row lie beside the next row's, so a staging instruction reads one contiguous
    1 KiB run; ``dense()`` gives the (rows, K) view for inspection."""
    __slots__ = ("hi", "lo", "scale", "rows", "cols")

    def __init__(self, rows, cols, device):
        assert cols % 32 == 0, "operand planes need K % 32 == 0"
        self.rows, self.cols = rows, cols
        self.hi = torch.empty(cols // 32, rows, 32, device=device, dtype=torch.float16)
        self.lo = torch.empty(cols // 32, rows, 32, device=device, dtype=torch.float16)
        self.scale = torch.empty(rows, device=device, dtype=torch.float32)

    def dense(self):
        """-> (hi, lo) as (rows, K) tensors (copies)"""
        return tuple(p_.permute(1, 0, 2).reshape(self.rows, self.cols) for p_ in (self.hi, self.lo))


def split_eligible(x, k_multiple=32):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 1 and x.is_contiguous() and x.shape[-1] % k_multiple == 0
            and x.data_ptr() % 16 == 0)


def split_rows(x, out=None, k_multiple=32):
    """x (..., K) contiguous fp32 -> SplitPlanes of its (R, K) view """
    _require(split_eligible(x, k_multiple), "split_rows: needs a contiguous fp32 CUDA tensor whose last dimension is a "
                                           f"multiple of {k_multiple}")
    K = x.shape[-1]
    R = x.numel() // K
    if out is None:
        out = SplitPlanes(R, K, x.device)
    _require(out.rows == R and out.cols == K, "split_rows: `out` has a different shape")
    with torch.cuda.device(x.device):
        rc = _lib.lib().gvl_split_rows_f16(x.data_ptr(), R, K, out.hi.data_ptr(), out.lo.data_ptr(), out.scale.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "split_rows")
    return out


def gemm_f16x3(a, b, bias=None, out=None):
    """a: SplitPlanes (R, K), b: SplitPlanes (N, K) -> a . b^T + bias, (R, N) fp32, at fp32 accuracy on the fp16 matrix
    cores (include/gvl_msda.h: gvl_gemm_f16x3_f32)"""
    _require(a.cols == b.cols, "gemm_f16x3: inner dimensions differ")
    _require(bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == b.rows),
             "gemm_f16x3: bias must be contiguous fp32 of length N")
    if out is None:
        out = torch.empty(a.rows, b.rows, device=a.hi.device, dtype=torch.float32)
    _require(out.dtype == torch.float32 and out.dim() == 2 and out.stride(1) == 1 and tuple(out.shape) == (a.rows, b.rows),
             "gemm_f16x3: `out` must be (R, N) fp32 with unit column stride")
    with torch.cuda.device(out.device):
        rc = _lib.lib().gvl_gemm_f16x3_f32(a.hi.data_ptr(), a.lo.data_ptr(), a.scale.data_ptr(), a.rows, b.hi.data_ptr(),
                                           b.lo.data_ptr(), b.scale.data_ptr(), b.rows, a.cols,
                                           bias.data_ptr() if bias is not None else None, out.data_ptr(), out.stride(0),
                                           torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "gemm_f16x3")
    return out


def ce_rows_forward(logits, target, weight):
    """(R, V) fp32 logits -> (out (R) = weight * (logits[r, target[r]] - logsumexp(logits[r])), lse (R)): the masked
    log-probability of the target word without the log-prob tensor (include/gvl_msda.h: gvl_ce_rows_forward_f32)"""
    _require(logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1,
             "ce_rows: logits must be an (R, V) fp32 CUDA matrix with unit column stride")
    R, V = logits.shape
    _require(target.dtype == torch.int64 and target.is_contiguous() and target.numel() == R and weight.dtype == torch.float32
             and weight.is_contiguous() and weight.numel() == R, "ce_rows: target int64 (R), weight fp32 (R)")
    out = torch.empty(R, device=logits.device, dtype=torch.float32)
    lse = torch.empty(R, device=logits.device, dtype=torch.float32)
    with torch.cuda.device(logits.device):
        rc = _lib.lib().gvl_ce_rows_forward_f32(logits.data_ptr(), logits.stride(0), R, V, target.data_ptr(),
                                                weight.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "ce_rows_forward")
    return out, lse


def ce_rows_backward_(logits, target, weight, grad_out, lse, amax=None):
    """overwrites `logits` with d sum(grad_out * out) / d logits (include/gvl_msda.h: gvl_ce_rows_backward_f32); amax (R) fp32
    receives an upper bound of every gradient row"""
    R, V = logits.shape
    _require(grad_out.dtype == torch.float32 and grad_out.is_contiguous() and grad_out.numel() == R,
             "ce_rows: grad_out fp32 (R)")
    with torch.cuda.device(logits.device):
        rc = _lib.lib().gvl_ce_rows_backward_f32(logits.data_ptr(), logits.stride(0), R, V, target.data_ptr(),
                                                 weight.data_ptr(), grad_out.data_ptr(), lse.data_ptr(),
                                                 amax.data_ptr() if amax is not None else None,
                                                 torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "ce_rows_backward")
    return logits


def col_sum_eligible(x):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.shape[0] > 0


def col_sum(x):
    """x (R, C) fp32 -> (C,) column sums (include/gvl_msda.h: gvl_col_sum_f32)."""
    _require(col_sum_eligible(x), "col_sum: x must be a 2-D fp32 CUDA matrix with unit column stride")
    out = torch.empty(x.shape[1], device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        rc = _lib.lib().gvl_col_sum_f32(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], out.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "col_sum")
    return out


def wgrad_eligible(dy, x):
    return (dy.is_cuda and dy.dtype == torch.float32 and x.dtype == torch.float32 and dy.dim() == 2 and x.dim() == 2
            and dy.shape[0] == x.shape[0] and dy.shape[0] > 0 and dy.stride(1) == 1 and x.stride(1) == 1
            and x.shape[1] % 4 == 0 and dy.stride(0) % 4 == 0 and x.stride(0) % 4 == 0
            and dy.stride(0) >= (dy.shape[1] + 3) // 4 * 4
            and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)


def wgrad(dy, x, amax_dy, amax_x, grad_w=None, grad_b=None, want_bias=True, accumulate=False, skip_zero_rows=False):
    """dy (R, N), x (R, K) fp32 -> (grad_w (N, K) = dy^T x, grad_b (N) = column sums of dy) on the fp16 matrix cores at fp32
    accuracy (include/gvl_msda.h: gvl_wgrad_f16x3_f32).  amax_*: fp32 vectors (or one number) whose maximum bounds |dy| / |x|.
    grad_w / grad_b given: written in place, with accumulate=True ON TOP of their contents.  skip_zero_rows (amax_dy per row):
    rows of dy whose bound is 0 -- all-zero rows -- are left out of the product (gvl_wgrad_f16x3_live_f32)."""
    _require(wgrad_eligible(dy, x), "wgrad: dy (R, N), x (R, K) fp32 CUDA matrices, unit column stride, N, K, strides % 4 == 0")
    R, N = dy.shape
    K = x.shape[1]
    for a_ in (amax_dy, amax_x):
        _require(a_.dtype == torch.float32 and a_.is_contiguous() and a_.numel() >= 1, "wgrad: amax must be contiguous fp32")
    if grad_w is None:
        _require(not accumulate, "wgrad: accumulate needs grad_w")
        grad_w = torch.empty(N, K, device=dy.device, dtype=torch.float32)
    if grad_b is None and want_bias:
        _require(not accumulate, "wgrad: accumulate needs grad_b")
        grad_b = torch.empty(N, device=dy.device, dtype=torch.float32)
    _require(grad_w.dtype == torch.float32 and grad_w.is_contiguous() and tuple(grad_w.shape) == (N, K)
             and (grad_b is None or (grad_b.dtype == torch.float32 and grad_b.is_contiguous() and grad_b.numel() == N)),
             "wgrad: grad_w (N, K) / grad_b (N) must be contiguous fp32")
    L = _lib.lib()
    nbytes = L.gvl_wgrad_workspace_bytes(R, N, K)
    ws = torch.empty(max(nbytes, 16) // 4, device=dy.device, dtype=torch.float32)
    live = None
    if skip_zero_rows and amax_dy.numel() == R:
        live = torch.empty(L.gvl_wgrad_live_ints(R), device=dy.device, dtype=torch.int32)
    with torch.cuda.device(dy.device):
        rc = L.gvl_wgrad_f16x3_live_f32(dy.data_ptr(), dy.stride(0), amax_dy.data_ptr(), amax_dy.numel(), x.data_ptr(), x.stride(0),
                                        amax_x.data_ptr(), amax_x.numel(), R, N, K, grad_w.data_ptr(),
                                        grad_b.data_ptr() if grad_b is not None else None, 1 if accumulate else 0,
                                        ws.data_ptr(), nbytes, live.data_ptr() if live is not None else None,
                                        torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "wgrad_f16x3")
    return grad_w, grad_b


class _WgradDesc(ctypes.Structure):                  # include/gvl_msda.h: gvl_wgrad_desc
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("amax_dy", ctypes.c_void_p), ("amax_x", ctypes.c_void_p),
                ("grad_w", ctypes.c_void_p), ("grad_b", ctypes.c_void_p), ("ld_dy", ctypes.c_int64), ("ld_x", ctypes.c_int64),
                ("n_amax_dy", ctypes.c_int), ("n_amax_x", ctypes.c_int), ("R", ctypes.c_int), ("N", ctypes.c_int),
                ("K", ctypes.c_int), ("accumulate", ctypes.c_int)]


def wgrad_group_max():
    return int(_lib.lib().gvl_wgrad_group_max())


def wgrad_group(items):
    """the weight / bias gradients of several Linears in ONE launch (+ one reduction): items = [(dy, x, amax_dy, amax_x, grad_w,
    grad_b | None)], at most wgrad_group_max() of them, every operand as wgrad() takes it; grad_w / grad_b are written in place
    (include/gvl_msda.h: gvl_wgrad_group_f16x3_f32)."""
    n = len(items)
    _require(0 < n <= wgrad_group_max(), "wgrad_group: 1 .. wgrad_group_max() problems")
    arr = (_WgradDesc * n)()
    for d, (dy, x, am_dy, am_x, gw, gb) in zip(arr, items):
        _require(wgrad_eligible(dy, x), "wgrad_group: dy (R, N), x (R, K) fp32 CUDA matrices, unit column stride, strides % 4 == 0")
        _require(gw.dtype == torch.float32 and gw.is_contiguous() and tuple(gw.shape) == (dy.shape[1], x.shape[1])
                 and (gb is None or (gb.dtype == torch.float32 and gb.is_contiguous() and gb.numel() == dy.shape[1])),
                 "wgrad_group: grad_w (N, K) / grad_b (N) must be contiguous fp32")
        d.dy, d.x, d.amax_dy, d.amax_x = dy.data_ptr(), x.data_ptr(), am_dy.data_ptr(), am_x.data_ptr()
        d.grad_w, d.grad_b = gw.data_ptr(), (gb.data_ptr() if gb is not None else None)
        d.ld_dy, d.ld_x, d.n_amax_dy, d.n_amax_x = dy.stride(0), x.stride(0), am_dy.numel(), am_x.numel()
        d.R, d.N, d.K, d.accumulate = dy.shape[0], dy.shape[1], x.shape[1], 0
    L = _lib.lib()
    dev = items[0][0].device
    nbytes = L.gvl_wgrad_group_workspace_bytes(ctypes.byref(arr), n)
    ws = torch.empty(max(nbytes, 16) // 4, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        rc = L.gvl_wgrad_group_f16x3_f32(ctypes.byref(arr), n, ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "wgrad_group_f16x3")


def lstm_cell_train_forward(gates_a, gates_b, gates_c, c_prev, act, h_out, c_out):
    n, H = c_prev.shape
    for name, t_ in (("gates_a", gates_a), ("gates_b", gates_b), ("gates_c", gates_c)):
        _f32_rows("lstm_cell_train: " + name, t_, 4 * H)
    for name, t_ in (("c_prev", c_prev), ("act", act), ("h_out", h_out), ("c_out", c_out)):
        _require(t_.is_cuda and t_.is_contiguous() and t_.dtype == torch.float32, f"lstm_cell_train: {name}")
    with torch.cuda.device(c_prev.device):
        rc = _lib.lib().gvl_lstm_cell_train_forward_f32(
            gates_a.data_ptr(), gates_a.stride(0), gates_b.data_ptr(), gates_b.stride(0), gates_c.data_ptr(),
            gates_c.stride(0), c_prev.data_ptr(), n, H, act.data_ptr(), h_out.data_ptr(), c_out.data_ptr(),
            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "lstm_cell_train_forward")


def lstm_cell_train_backward(grad_h_a, grad_h_b, grad_c, act, c_prev, c_new, grad_gates, grad_c_prev, gates_sum=None, first=False):
    """gates_sum (n, 4H) contiguous: the running sum of grad_gates over the steps (first: this call starts it)"""
    n, H = c_prev.shape
    _f32_rows("lstm_cell_train: grad_gates", grad_gates, 4 * H)
    for name, t_ in (("grad_h_a", grad_h_a), ("grad_h_b", grad_h_b), ("grad_c", grad_c), ("act", act),
                     ("c_prev", c_prev), ("c_new", c_new), ("grad_c_prev", grad_c_prev)):
        _require(t_ is None or (t_.is_cuda and t_.is_contiguous() and t_.dtype == torch.float32),
                 f"lstm_cell_train: {name} must be a contiguous fp32 CUDA tensor")
    ptr = lambda t_: None if t_ is None else t_.data_ptr()      # noqa: E731
    with torch.cuda.device(c_prev.device):
        _require(gates_sum is None or (gates_sum.is_cuda and gates_sum.is_contiguous() and gates_sum.dtype == torch.float32
                                       and tuple(gates_sum.shape) == (n, 4 * H)), "lstm_cell_train: gates_sum must be (n, 4H) fp32")
        rc = _lib.lib().gvl_lstm_cell_train_backward_sum_f32(
            ptr(grad_h_a), ptr(grad_h_b), ptr(grad_c), act.data_ptr(), c_prev.data_ptr(), c_new.data_ptr(), n, H,
            grad_gates.data_ptr(), grad_gates.stride(0), grad_c_prev.data_ptr(), ptr(gates_sum), int(bool(first)),
            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "lstm_cell_train_backward")


def lstm_cell(gates_a, gates_b, emb_gates, it, c, gates_c=None, planes=False):
    """fused LSTM cell pointwise step (include/gvl_msda.h: gvl_lstm_cell_f32 / _bf16) -> (h', c').  The gate operands
    are all fp32 or all bf16; the state c (and the outputs) fp32.  planes=True (fp32 gates): h' additionally as the
    SplitPlanes operand of gemm_f16x3, left in h'._gvl_planes (gvl_lstm_cell_split_f32)."""
    n, H = c.shape
    gt = gates_a.dtype
    _require(gt in (torch.float32, torch.bfloat16) and c.dtype == torch.float32, "lstm_cell: gates fp32 | bf16, c fp32")
    for name, t_ in (("gates_a", gates_a), ("gates_b", gates_b)) + ((("gates_c", gates_c),) if gates_c is not None else ()):
        _require(t_.is_cuda and t_.dtype == gt and t_.dim() == 2 and t_.stride(1) == 1
                 and t_.shape == (n, 4 * H), f"lstm_cell: {name} must be an (n, 4H) CUDA matrix of the gates' dtype")
    _require(emb_gates.is_contiguous() and emb_gates.dtype == gt and c.is_contiguous() and it.is_contiguous()
             and it.dtype == torch.int64, "lstm_cell: emb_gates / c / it must be contiguous (it int64)")
    h_out, c_out = torch.empty_like(c), torch.empty_like(c)
    if planes:
        _require(gt == torch.float32 and H % 32 == 0, "lstm_cell: planes=True needs fp32 gates and H % 32 == 0")
        hp = SplitPlanes(n, H, c.device)
        with torch.cuda.device(c.device):
            rc = _lib.lib().gvl_lstm_cell_split_f32(
                gates_a.data_ptr(), gates_a.stride(0), gates_b.data_ptr(), gates_b.stride(0), emb_gates.data_ptr(),
                it.data_ptr(), gates_c.data_ptr() if gates_c is not None else None,
                gates_c.stride(0) if gates_c is not None else 0, c.data_ptr(), n, H, h_out.data_ptr(), c_out.data_ptr(),
                hp.hi.data_ptr(), hp.lo.data_ptr(), hp.scale.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "lstm_cell_split")
        h_out._gvl_planes = hp
        return h_out, c_out
    extra = ()
    if gt == torch.bfloat16:                      # a bf16 copy of h' for the next GEMMs (returned as h_out._gvl_lowp)
        h_lp = torch.empty((n, H), dtype=gt, device=c.device)
        extra = (h_lp.data_ptr(),)
    with torch.cuda.device(c.device):
        rc = getattr(_lib.lib(), "gvl_lstm_cell_" + _SUFFIX[gt])(
            gates_a.data_ptr(), gates_a.stride(0), gates_b.data_ptr(), gates_b.stride(0), emb_gates.data_ptr(),
            it.data_ptr(), gates_c.data_ptr() if gates_c is not None else None,
            gates_c.stride(0) if gates_c is not None else 0, c.data_ptr(), n, H, h_out.data_ptr(), c_out.data_ptr(),
            *extra, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "lstm_cell")
    if extra:
        h_out._gvl_lowp = h_lp
    return h_out, c_out


def gate_permutation(H, device=None):
    """row index of nn.LSTM's (4H, .) gate-major weights in the order 4 * unit + gate (gvl_gemm_f16x3_lstm_f32)"""
    return torch.arange(4 * H, device=device).view(4, H).t().reshape(-1)


def gemm_f16x3_lstm(a, w, gates_h, gates_c, emb_gates, it, c):
    """(h', c') = LSTM cell of a . w^T + gates_c + gates_h + emb_gates[it] (include/gvl_msda.h: gvl_gemm_f16x3_lstm_f32):
    a SplitPlanes (n, K), w SplitPlanes (4H, K) with rows in gate_permutation order, gates_h / gates_c (n, 4H) fp32 (unit
    column stride) and emb_gates (V + 1, 4H) with columns in the same order; h' carries its planes (h'._gvl_planes)."""
    n, H = c.shape
    _require(a.rows == n and w.rows == 4 * H and a.cols == w.cols, "gemm_f16x3_lstm: operand shapes")
    for name, t_ in (("gates_h", gates_h),) + ((("gates_c", gates_c),) if gates_c is not None else ()):
        _require(t_.is_cuda and t_.dtype == torch.float32 and t_.dim() == 2 and t_.stride(1) == 1
                 and tuple(t_.shape) == (n, 4 * H), f"gemm_f16x3_lstm: {name} must be an (n, 4H) fp32 CUDA matrix")
    _require(emb_gates.is_contiguous() and emb_gates.dtype == torch.float32 and emb_gates.shape[1] == 4 * H
             and c.is_contiguous() and c.dtype == torch.float32 and it.is_contiguous() and it.dtype == torch.int64
             and it.numel() == n, "gemm_f16x3_lstm: emb_gates (V+1, 4H) / c fp32 contiguous, it int64 (n)")
    h_out, c_out = torch.empty_like(c), torch.empty_like(c)
    hp = SplitPlanes(n, H, c.device)
    with torch.cuda.device(c.device):
        rc = _lib.lib().gvl_gemm_f16x3_lstm_f32(
            a.hi.data_ptr(), a.lo.data_ptr(), a.scale.data_ptr(), n, w.hi.data_ptr(), w.lo.data_ptr(), w.scale.data_ptr(),
            H, a.cols, gates_h.data_ptr(), gates_h.stride(0), gates_c.data_ptr() if gates_c is not None else None,
            gates_c.stride(0) if gates_c is not None else 0, emb_gates.data_ptr(), it.data_ptr(), c.data_ptr(),
            h_out.data_ptr(), c_out.data_ptr(), hp.hi.data_ptr(), hp.lo.data_ptr(), hp.scale.data_ptr(),
            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "gemm_f16x3_lstm")
    h_out._gvl_planes = hp
    return h_out, c_out


def f16_products_now():
    """3 | 1: the number of fp16 products per fp32 product the library currently runs (gvl_f16_products(0))"""
    return int(_lib.lib().gvl_f16_products(0))


def gates_applicable(n, H):
    """whether gvl_gemm_f16x3_gates_f32's tiles fill the chip at (n, 4H) (else the two-launch form is the faster one)"""
    return bool(_lib.lib().gvl_gemm_f16x3_gates_applicable(n, H))


def gemm_f16x3_gates(a, h_prev, w, gates_c, emb_gates, it, c, need_h=True):
    """(h', c') = LSTM cell of [h_prev | a] . w^T + gates_c + emb_gates[it] (include/gvl_msda.h: gvl_gemm_f16x3_gates_f32):
    a SplitPlanes (n, K_a), h_prev SplitPlanes (n, K_h) -- the planes of the step's incoming hidden state --, w SplitPlanes
    (4H, K_h + K_a) = [W_hh | W_ih[:, attention columns]] with rows in gate_permutation order; gates_c (n, 4H) / emb_gates
    (V + 1, 4H) with columns in that order.  h' carries its planes; need_h=False: ONLY its planes are written (the returned h'
    is an unwritten buffer marked ``_gvl_planes_only``)."""
    n, H = c.shape
    _require(a.rows == n and h_prev.rows == n and w.rows == 4 * H and a.cols + h_prev.cols == w.cols,
             "gemm_f16x3_gates: operand shapes")
    if gates_c is not None:
        _require(gates_c.is_cuda and gates_c.dtype == torch.float32 and gates_c.dim() == 2 and gates_c.stride(1) == 1
                 and tuple(gates_c.shape) == (n, 4 * H), "gemm_f16x3_gates: gates_c must be an (n, 4H) fp32 CUDA matrix")
    _require(emb_gates.is_contiguous() and emb_gates.dtype == torch.float32 and emb_gates.shape[1] == 4 * H
             and c.is_contiguous() and c.dtype == torch.float32 and it.is_contiguous() and it.dtype == torch.int64
             and it.numel() == n, "gemm_f16x3_gates: emb_gates (V+1, 4H) / c fp32 contiguous, it int64 (n)")
    h_out, c_out = torch.empty_like(c), torch.empty_like(c)
    hp = SplitPlanes(n, H, c.device)
    with torch.cuda.device(c.device):
        rc = _lib.lib().gvl_gemm_f16x3_gates_f32(
            a.hi.data_ptr(), a.lo.data_ptr(), a.scale.data_ptr(), h_prev.hi.data_ptr(), h_prev.lo.data_ptr(),
            h_prev.scale.data_ptr(), n, w.hi.data_ptr(), w.lo.data_ptr(), w.scale.data_ptr(), H, h_prev.cols, a.cols,
            gates_c.data_ptr() if gates_c is not None else None, gates_c.stride(0) if gates_c is not None else 0,
            emb_gates.data_ptr(), it.data_ptr(), c.data_ptr(), h_out.data_ptr() if need_h else None, c_out.data_ptr(), hp.hi.data_ptr(),
            hp.lo.data_ptr(), hp.scale.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "gemm_f16x3_gates")
    h_out._gvl_planes = hp
    if not need_h:
        h_out._gvl_planes_only = True
    return h_out, c_out


def row_argmax_lse(logits):
    """(R, V) fp32 | bf16 -> (argmax int64 (R,), log_softmax value at the argmax (R,)); first maximal index on ties."""
    _require(logits.is_cuda and logits.is_contiguous() and logits.dtype in (torch.float32, torch.bfloat16)
             and logits.dim() == 2, "row_argmax_lse: logits must be a contiguous fp32 / bf16 CUDA matrix")
    R, V = logits.shape
    idx = torch.empty(R, dtype=torch.int64, device=logits.device)
    lp = torch.empty(R, dtype=torch.float32, device=logits.device)
    with torch.cuda.device(logits.device):
        rc = getattr(_lib.lib(), "gvl_row_argmax_lse_" + _SUFFIX[logits.dtype])(logits.data_ptr(), R, V, idx.data_ptr(), lp.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "row_argmax_lse")
    return idx, lp


class GreedyPartials:
    """what gvl_gemm_f16x3_argmax_f32 leaves instead of the (R, V) logits: per row and 64 vocabulary entries
    {max, sum exp(v - max), index}"""
    __slots__ = ("part", "rows", "vocab")

    def __init__(self, part, rows, vocab):
        self.part, self.rows, self.vocab = part, rows, vocab


def gemm_f16x3_argmax(x, w, bias=None):
    """x: SplitPlanes (R, K) hidden states, w: SplitPlanes (V, K) vocabulary weight -> GreedyPartials of x . w^T + bias
    (include/gvl_msda.h: gvl_gemm_f16x3_argmax_f32); consumed by greedy_step / row_argmax_lse_partials"""
    _require(x.cols == w.cols, "gemm_f16x3_argmax: inner dimensions differ")
    _require(bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == w.rows),
             "gemm_f16x3_argmax: bias must be contiguous fp32 of length V")
    L = _lib.lib()
    part = torch.empty(L.gvl_gemm_f16x3_argmax_chunks(w.rows), x.rows, 4, device=x.hi.device, dtype=torch.float32)
    with torch.cuda.device(part.device):
        rc = L.gvl_gemm_f16x3_argmax_f32(x.hi.data_ptr(), x.lo.data_ptr(), x.scale.data_ptr(), x.rows, w.hi.data_ptr(),
                                         w.lo.data_ptr(), w.scale.data_ptr(), w.rows, x.cols,
                                         bias.data_ptr() if bias is not None else None, part.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "gemm_f16x3_argmax")
    return GreedyPartials(part, x.rows, w.rows)


def _greedy_from_partials(p, first, unfinished, seq_ptr, lp_ptr, T, alive_ptr=None):
    tok = torch.empty(p.rows, dtype=torch.int64, device=p.part.device)
    lp = torch.empty(p.rows, dtype=torch.float32, device=p.part.device)
    with torch.cuda.device(p.part.device):
        rc = _lib.lib().gvl_greedy_step_partials_alive_f32(p.part.data_ptr(), p.rows, p.vocab, first, tok.data_ptr(),
                                                           lp.data_ptr(), unfinished, seq_ptr, lp_ptr, T, alive_ptr,
                                                           torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "greedy_step_partials")
    return tok, lp


def row_argmax_lse_partials(p):
    """GreedyPartials -> (argmax (R,) int64, log-softmax at the argmax (R,) fp32)"""
    return _greedy_from_partials(p, 0, None, None, None, 0)


def greedy_step(logits, t_col, unfinished, seq, seq_lp, alive=None):
    """argmax + log-softmax-at-argmax of the (R, V) logits AND the greedy bookkeeping of decoding step t_col
    (include/gvl_msda.h: gvl_greedy_step_f32): updates unfinished (R,) uint8 and seq / seq_lp (R, T) in place at
    column t_col; -> raw argmax tokens (R,) int64.  `logits` may be the GreedyPartials of gemm_f16x3_argmax.
    alive (T,) uint8, zero-initialised: alive[t_col] is set when any row is still unfinished after the step (the loop-exit
    test of LSTM_DSA.py:186-187) -- by the kernel itself on the partials path, by two small ops otherwise."""
    if isinstance(logits, GreedyPartials):
        _require(unfinished.dtype == torch.uint8 and seq.dtype == torch.int64 and seq_lp.dtype == torch.float32
                 and seq.is_contiguous() and seq_lp.is_contiguous() and seq.shape == seq_lp.shape
                 and seq.shape[0] == logits.rows, "greedy_step: bad bookkeeping tensors")
        _require(alive is None or (alive.dtype == torch.uint8 and alive.is_contiguous() and alive.numel() > t_col),
                 "greedy_step: alive must be a (T,) uint8 tensor")
        return _greedy_from_partials(logits, 1 if t_col == 0 else 0, unfinished.data_ptr(), seq.data_ptr() + 8 * t_col,
                                     seq_lp.data_ptr() + 4 * t_col, seq.shape[1],
                                     alive.data_ptr() + t_col if alive is not None else None)[0]
    _require(logits.is_cuda and logits.is_contiguous() and logits.dtype in (torch.float32, torch.bfloat16)
             and logits.dim() == 2, "greedy_step: logits must be a contiguous fp32 / bf16 CUDA matrix")
    R, V = logits.shape
    _require(unfinished.dtype == torch.uint8 and seq.dtype == torch.int64
             and seq_lp.dtype == torch.float32 and seq.is_contiguous() and seq_lp.is_contiguous()
             and seq.shape == seq_lp.shape and seq.shape[0] == R, "greedy_step: bad bookkeeping tensors")
    T = seq.shape[1]
    tok = torch.empty(R, dtype=torch.int64, device=logits.device)
    lp = torch.empty(R, dtype=torch.float32, device=logits.device)
    with torch.cuda.device(logits.device):
        rc = getattr(_lib.lib(), "gvl_greedy_step_" + _SUFFIX[logits.dtype])(
            logits.data_ptr(), R, V, 1 if t_col == 0 else 0, tok.data_ptr(), lp.data_ptr(), unfinished.data_ptr(),
            seq.data_ptr() + 8 * t_col, seq_lp.data_ptr() + 4 * t_col, T,
            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "greedy_step")
    if alive is not None:
        alive[t_col] = (seq[:, t_col] != 0).any()
    return tok


def greedy_step_and_gemm(logits, t_col, unfinished, seq, seq_lp, alive, a, b, bias=None):
    """greedy_step on GreedyPartials AND, in the same launch, gemm_f16x3(a, b, bias) (include/gvl_msda.h:
    gvl_greedy_step_partials_gemm_f32) -> (raw argmax tokens (R,) int64, a . b^T + bias (Ra, Nb) fp32)"""
    _require(isinstance(logits, GreedyPartials), "greedy_step_and_gemm: logits must be the GreedyPartials of gemm_f16x3_argmax")
    _require(unfinished.dtype == torch.uint8 and seq.dtype == torch.int64 and seq_lp.dtype == torch.float32
             and seq.is_contiguous() and seq_lp.is_contiguous() and seq.shape == seq_lp.shape
             and seq.shape[0] == logits.rows, "greedy_step_and_gemm: bad bookkeeping tensors")
    _require(alive is None or (alive.dtype == torch.uint8 and alive.is_contiguous() and alive.numel() > t_col),
             "greedy_step_and_gemm: alive must be a (T,) uint8 tensor")
    _require(a.cols == b.cols, "greedy_step_and_gemm: inner dimensions differ")
    _require(bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == b.rows),
             "greedy_step_and_gemm: bias must be contiguous fp32 of length N")
    p = logits
    tok = torch.empty(p.rows, dtype=torch.int64, device=p.part.device)
    lp = torch.empty(p.rows, dtype=torch.float32, device=p.part.device)
    out = torch.empty(a.rows, b.rows, device=a.hi.device, dtype=torch.float32)
    with torch.cuda.device(p.part.device):
        rc = _lib.lib().gvl_greedy_step_partials_gemm_f32(
            p.part.data_ptr(), p.rows, p.vocab, 1 if t_col == 0 else 0, tok.data_ptr(), lp.data_ptr(), unfinished.data_ptr(),
            seq.data_ptr() + 8 * t_col, seq_lp.data_ptr() + 4 * t_col, seq.shape[1],
            alive.data_ptr() + t_col if alive is not None else None, a.hi.data_ptr(), a.lo.data_ptr(), a.scale.data_ptr(), a.rows,
            b.hi.data_ptr(), b.lo.data_ptr(), b.scale.data_ptr(), b.rows, a.cols, bias.data_ptr() if bias is not None else None,
            out.data_ptr(), out.stride(0), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "greedy_step_partials_gemm")
    return tok, out


PROF_TAGS = {1: "fwd_t1d_d64", 2: "fwd_generic", 3: "bwd_t1d_d64", 4: "bwd_generic", 5: "sample", 6: "sum_partials",
             7: "sample_bwd", 8: "cap_attend", 9: "row_argmax_lse", 10: "lstm_cell", 11: "lsap",
             12: "cap_train_fwd", 13: "cap_train_bwd", 14: "lstm_train", 15: "match_cost", 16: "criterion",
             17: "pos_embed", 18: "col_sum", 19: "proj", 20: "split_rows", 21: "gemm_f16x3", 22: "layer_gemm",
             23: "layer_norm_etc", 24: "wgrad_f16x3", 25: "mha_train"}


def clock_probe_mhz(device=None, n_fma=20000):
    """the shader clock the chip holds right now on the current stream (include/gvl_msda.h: gvl_clock_probe): a short chain
    of dependent FMAs timed with the cycle counter against the 100 MHz wall clock; synchronises"""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    out = torch.zeros(2, dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        rc = _lib.lib().gvl_clock_probe(out.data_ptr(), int(n_fma), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "clock_probe")
    cyc, ref = out.tolist()
    return 100.0 * cyc / max(ref, 1)


class f16_products:
    """``with f16_products(1):`` -- every split-fp16 product launched inside (token-loop GEMMs, the inference layers' Linear
    kernel) spends ONE fp16 product per fp32 product instead of three: operands rounded to fp16 at their row scale (11
    significant bits against bf16's 8), fp32 accumulation (include/gvl_msda.h: gvl_f16_products).  What inference under
    torch.autocast runs on (gvl_amd/pdvc.py: autocast_inference_policy).  Thread-local, restored on exit; a hipGraph captured
    inside keeps the kernels that were selected at capture."""

    def __init__(self, n):
        self.n, self.prev = int(n), None

    def __enter__(self):
        self.prev = _lib.lib().gvl_f16_products(self.n)
        _lib.check(min(self.prev, 0), "f16_products")
        return self

    def __exit__(self, *exc):
        _lib.lib().gvl_f16_products(self.prev)
        return False


def keeps_products(cls):
    """Class decorator for an autograd Function whose BACKWARD launches split-fp16 products: the backward runs with the product
    count its forward ran with.  ``f16_products`` is thread-local and the backward runs later, on autograd's device thread --
    without this a training step under the autocast policy (gvl_amd/pdvc.py: one product per fp32 product) would take its input-
    and weight-gradient products at three."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args, **kwargs):
        ctx._gvl_products = f16_products_now()
        return fwd(ctx, *args, **kwargs)

    def backward(ctx, *grads):
        n = getattr(ctx, "_gvl_products", 3)
        if n == f16_products_now():
            return bwd(ctx, *grads)
        with f16_products(n):
            return bwd(ctx, *grads)
    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


def profile_enable(on=True):
    """on: False / 0 = off; True / 1 = per-dispatch stamps of the sampling-path kernels; 2 = additionally the projection
    kernel in front of them (stamping two consecutive launches inflates the second one's interval, see gvl_common.hpp)"""
    _lib.lib().gvl_prof_enable(int(on))


def profile_collect(capacity=1 << 16):
    """-> list of (kernel tag name, meta_a, meta_b, microseconds) for every launch since the last collect."""
    us = np.empty(capacity, np.float32)
    tag = np.empty(capacity, np.int32)
    ma = np.empty(capacity, np.int32)
    mb = np.empty(capacity, np.int32)
    n = _lib.lib().gvl_prof_collect(_hp(us), _hp(tag), _hp(ma), _hp(mb), capacity)
    return [(PROF_TAGS.get(int(tag[i]), str(tag[i])), int(ma[i]), int(mb[i]), float(us[i])) for i in range(n)]
