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
_SUFFIX = {torch.float32: "f32", torch.float64: "f64"}


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
    _require(value.dtype in _SUFFIX, f"ms_deform_attn: unsupported dtype {value.dtype} (fp32 / fp64 only)")
    _require(spatial_shapes.dtype == torch.int64 and level_start_index.dtype == torch.int64,
             "spatial_shapes / level_start_index must be int64")
    B, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Q, P = sampling_loc.shape[1], sampling_loc.shape[4]
    _require(tuple(sampling_loc.shape) == (B, Q, M, L, P, 2), "sampling_loc has wrong shape")
    if attn_weight is not None:
        _require(tuple(attn_weight.shape) == (B, Q, M, L, P), "attn_weight has wrong shape")
        _require(attn_weight.dtype == value.dtype, "dtype mismatch")
    _require(sampling_loc.dtype == value.dtype, "dtype mismatch")
    step = min(B, int(im2col_step)) if B > 0 else 1
    _require(step > 0 and B % step == 0, f"batch({B}) must divide im2col_step({step})")   # cu:50-52
    return B, S, M, D, L, Q, P


def _hp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


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
    _lib.check(rc, "ms_deform_attn_forward")
    return out


def ms_deform_attn_sample(value, spatial_shapes, level_start_index, sampling_loc, pad_mode="border"):
    """ms_deform_attn_core_pytorch(..., return_value=True) (func.py:67-68): (B*M, D, Q, L, P)."""
    B, S, M, D, L, Q, P = _common_checks(value, spatial_shapes, level_start_index, sampling_loc, None, 1 << 30)
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
    sh, ls = host_shapes(spatial_shapes, level_start_index)
    grad_value = torch.empty_like(value)
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    lib = _lib.lib()
    nbytes = lib.gvl_msda_backward_workspace_bytes(B, S, M, D, L, Q, P, value.element_size())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=value.device) if nbytes else None
    fn = getattr(lib, "gvl_msda_backward_" + _SUFFIX[value.dtype])
    with torch.cuda.device(value.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = fn(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                attn_weight.data_ptr(), grad_output.data_ptr(), B, S, M, D, L, Q, P, PAD_MODES[pad_mode], _hp(sh),
                _hp(ls), grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(),
                ws.data_ptr() if ws is not None else None, nbytes, stream)
    _lib.check(rc, "ms_deform_attn_backward")
    return grad_value, grad_loc, grad_attn
