"""ctypes/numpy front-end of oracle/msda_ref.c (TEST INFRASTRUCTURE ONLY).

Function-for-function restatement targets (in /root/reference):
  msda_forward(pad_mode="zeros")   MSDeformAttnFunction.forward  pdvc/ops/functions/ms_deform_attn_func.py:25-31
                                   -> ms_deform_attn_cuda_forward pdvc/ops/src/cuda/ms_deform_attn_cuda.cu:20-80
  msda_backward(pad_mode="zeros")  MSDeformAttnFunction.backward func.py:33-41 -> cu:83-153
  msda_forward(pad_mode="border")  ms_deform_attn_core_pytorch   func.py:44-71
  msda_sample                      ms_deform_attn_core_pytorch(return_value=True) func.py:67-68
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
PAD = {"zeros": 0, "border": 1}


def build(force=False):
    so = os.path.join(_HERE, "libgvl_oracle.so")
    src = os.path.join(_HERE, "msda_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libgvl_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, lsi, loc, aw=None):
    dt = value.dtype
    assert dt in (np.float32, np.float64)
    value = np.ascontiguousarray(value)
    loc = np.ascontiguousarray(loc, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    if shapes.ndim == 1:  # temporal shapes (T_l,) -> (1, T_l)  (ms_deform_attn.py:114-117)
        shapes = np.stack([np.ones_like(shapes), shapes], -1)
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    B, S, M, D = value.shape
    _, Q, _, L, P, two = loc.shape
    assert two == 2 and shapes.shape == (L, 2) and lsi.shape == (L,)
    assert int((shapes[:, 0] * shapes[:, 1]).sum()) == S
    if aw is not None:
        aw = np.ascontiguousarray(aw, dtype=dt)
        assert aw.shape == (B, Q, M, L, P)
    suf = "f32" if dt == np.float32 else "f64"
    return value, shapes, lsi, loc, aw, (B, S, M, D, L, Q, P), suf


def msda_forward(value, shapes, lsi, loc, aw, pad_mode="zeros"):
    value, shapes, lsi, loc, aw, dims, suf = _prep(value, shapes, lsi, loc, aw)
    B, S, M, D, L, Q, P = dims
    out = np.empty((B, Q, M * D), dtype=value.dtype)
    rc = getattr(lib(), f"oracle_msda_fwd_{suf}")(
        _p(value), _p(shapes), _p(lsi), _p(loc), _p(aw), B, S, M, D, L, Q, P, PAD[pad_mode], _p(out))
    assert rc == 0
    return out


def msda_sample(value, shapes, lsi, loc, pad_mode="border"):
    value, shapes, lsi, loc, _, dims, suf = _prep(value, shapes, lsi, loc)
    B, S, M, D, L, Q, P = dims
    out = np.empty((B * M, D, Q, L, P), dtype=value.dtype)
    rc = getattr(lib(), f"oracle_msda_sample_{suf}")(
        _p(value), _p(shapes), _p(lsi), _p(loc), B, S, M, D, L, Q, P, PAD[pad_mode], _p(out))
    assert rc == 0
    return out


def msda_backward(value, shapes, lsi, loc, aw, gout, pad_mode="zeros"):
    value, shapes, lsi, loc, aw, dims, suf = _prep(value, shapes, lsi, loc, aw)
    B, S, M, D, L, Q, P = dims
    gout = np.ascontiguousarray(gout, dtype=value.dtype).reshape(B, Q, M * D)
    gv = np.empty_like(value)
    gl = np.empty_like(loc)
    gw = np.empty_like(aw)
    rc = getattr(lib(), f"oracle_msda_bwd_{suf}")(
        _p(value), _p(shapes), _p(lsi), _p(loc), _p(aw), _p(gout), B, S, M, D, L, Q, P,
        PAD[pad_mode], _p(gv), _p(gl), _p(gw))
    assert rc == 0
    return gv, gl, gw
