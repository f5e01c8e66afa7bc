"""the reference keeps ALL of its Python and only the native extension module is replaced --
``sys.modules["MultiScaleDeformableAttention"] = gvl_amd.MultiScaleDeformableAttention`` -- so that the ``import
MultiScaleDeformableAttention as MSDA`` at ms_deform_attn_func.py:18-21 resolves to the C-ABI shim.  The autograd
Function below has the reference Function's shape (ms_deform_attn_func.py:23-41: forward through
MSDA.ms_deform_attn_forward, saved tensors, once-differentiable backward through MSDA.ms_deform_attn_backward, a
6-tuple with None for the non-tensor inputs) but imports the extension BY ITS REFERENCE NAME; tensors are built the way
the reference's modules build them (plain torch tensors, no host-side shape cache attached)."""
import importlib
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def reference_named_extension():
    import gvl_amd.MultiScaleDeformableAttention as shim
    saved = sys.modules.get("MultiScaleDeformableAttention")
    sys.modules["MultiScaleDeformableAttention"] = shim
    try:
        yield importlib.import_module("MultiScaleDeformableAttention")
    finally:
        if saved is None:
            sys.modules.pop("MultiScaleDeformableAttention", None)
        else:
            sys.modules["MultiScaleDeformableAttention"] = saved


def make_function():
    import MultiScaleDeformableAttention as MSDA          # resolved through sys.modules, as in the reference

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, value, shapes, lsi, loc, attn, im2col_step):
            ctx.im2col_step = im2col_step
            ctx.save_for_backward(value, shapes, lsi, loc, attn)
            return MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, attn, im2col_step)

        @staticmethod
        @torch.autograd.function.once_differentiable
        def backward(ctx, grad_output):
            value, shapes, lsi, loc, attn = ctx.saved_tensors
            gv, gl, ga = MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, attn, grad_output.contiguous(),
                                                      ctx.im2col_step)
            return gv, None, None, gl, ga, None
    return Fn, MSDA


def inputs(B, Q, lens, seed=0, M=8, D=64, P=4):
    rs = np.random.RandomState(seed)
    L, S = len(lens), sum(lens)
    shapes = np.array([(1, x) for x in lens], np.int64)
    lsi = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    value = rs.standard_normal((B, S, M, D)).astype(np.float32)
    loc = rs.uniform(-0.25, 1.25, (B, Q, M, L, P, 2)).astype(np.float32)
    loc[..., 1] = 0.5
    aw = rs.rand(B, Q, M, L, P).astype(np.float32)
    aw /= aw.sum((-1, -2), keepdims=True)
    gout = rs.standard_normal((B, Q, M * D)).astype(np.float32)
    return value, shapes, lsi, loc, aw, gout


def test_reference_shaped_function_over_the_swapped_extension(reference_named_extension):
    from oracle import msda_oracle as O
    Fn, MSDA = make_function()
    assert MSDA is reference_named_extension
    value, shapes, lsi, loc, aw, gout = inputs(2, 37, [20, 10, 5, 3])
    tt = lambda a: torch.from_numpy(a).to(DEV)                                   # noqa: E731
    v, l_, a_ = tt(value).requires_grad_(), tt(loc).requires_grad_(), tt(aw).requires_grad_()
    out = Fn.apply(v, tt(shapes), tt(lsi), l_, a_, 64)
    out.backward(tt(gout))
    ref = O.msda_forward(value, shapes, lsi, loc, aw, "zeros")
    rv, rl, rw = O.msda_backward(value, shapes, lsi, loc, aw, gout, "zeros")
    assert np.abs(out.detach().cpu().numpy() - ref).max() < 1e-4
    assert np.abs(v.grad.cpu().numpy() - rv).max() < 1e-4
    assert np.abs(l_.grad.cpu().numpy() - rl).max() < 1e-4 * max(1.0, np.abs(rl).max())
    assert np.abs(a_.grad.cpu().numpy() - rw).max() < 1e-4 * max(1.0, np.abs(rw).max())
    from gvl_amd import _lib
    assert _lib.lib().gvl_msda_last_impl() == 2                                   # the temporal fast kernels ran
    # the reference's error behaviour through the same route (ms_deform_attn_cuda.cu:28-52, ms_deform_attn.h:38)
    with pytest.raises(RuntimeError, match="contiguous"):
        MSDA.ms_deform_attn_forward(tt(value).transpose(1, 2), tt(shapes), tt(lsi), tt(loc), tt(aw), 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(torch.from_numpy(value), torch.from_numpy(shapes), torch.from_numpy(lsi),
                                    torch.from_numpy(loc), torch.from_numpy(aw), 64)
    with pytest.raises(RuntimeError, match="im2col_step"):
        MSDA.ms_deform_attn_forward(tt(np.concatenate([value, value[:1]])), tt(shapes), tt(lsi),           # B = 3
                                    tt(np.concatenate([loc, loc[:1]])), tt(np.concatenate([aw, aw[:1]])), 2)


def test_unfused_decoder_call_has_no_hidden_host_cost(reference_named_extension):
    """the decoder-shaped call through this route must cost tens of microseconds of host time, not
    ~800 (the r01 kernel sweep's figure was an artefact of that tool; measured 15 us per call: tools/dbg_host_cost.py).
    The one device->host read (level shapes, to pick the temporal kernels) happens on the FIRST call per shapes tensor."""
    Fn, MSDA = make_function()
    value, shapes, lsi, loc, aw, _ = inputs(16, 300, [100, 50, 25, 13])
    tt = lambda a: torch.from_numpy(a).to(DEV)                                   # noqa: E731
    v, sh, ls, lo, a_ = tt(value), tt(shapes), tt(lsi), tt(loc), tt(aw)
    for _ in range(5):
        MSDA.ms_deform_attn_forward(v, sh, ls, lo, a_, 64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        MSDA.ms_deform_attn_forward(v, sh, ls, lo, a_, 64)
    torch.cuda.synchronize()
    per_call = (time.perf_counter() - t0) / 100 * 1e6
    assert per_call < 150.0, per_call


def test_the_ctypes_binding_printed_in_integration_md_works_as_written():
    """INTEGRATION.md section 1 prints a ~15-line ctypes binding of gvl_msda_forward_f32; run exactly that text (library
    path made absolute) against the oracle, so the document cannot drift from the ABI."""
    import os
    import re
    from oracle import msda_oracle as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = next(b for b in blocks if "gvl_msda_forward_f32.argtypes" in b)
    code = code.replace('"gvl_amd/libgvl_msda.so"', repr(os.path.join(root, "gvl_amd", "libgvl_msda.so")))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    value, shapes, lsi, loc, aw, _ = inputs(2, 19, [12, 6, 3, 2], seed=4)
    tt = lambda a: torch.from_numpy(a).to(DEV)                                   # noqa: E731
    out = ns["ms_deform_attn_forward"](tt(value), tt(shapes), tt(lsi), tt(loc), tt(aw), 64)
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - O.msda_forward(value, shapes, lsi, loc, aw, "zeros")).max() < 1e-4
