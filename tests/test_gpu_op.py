"""GPU parity tests of the op, through the C ABI (ctypes shim gvl_amd.MultiScaleDeformableAttention):
  * against the golden vectors produced by the imported reference (tests/golden/op_*.npz),
  * against the CPU oracle (oracle/msda_ref.c) on seeded inputs up to BASELINE.json's full size,
  * size-independent properties at full size (linearity, generic == fast, run-to-run bitwise determinism).
Tolerances: fp64 1e-10; fp32 outputs 1e-4 absolute (north_star), gradients 1e-4 relative to their scale."""
import glob
import zlib
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load, t, maxerr, level_lengths

pytestmark = pytest.mark.gpu

OP_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "op_*.npz")))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def MSDA():
    from gvl_amd import MultiScaleDeformableAttention as m, _lib
    _lib.lib()
    return m


def set_impl(name):
    from gvl_amd import _lib
    _lib.lib().gvl_msda_set_impl({"auto": 0, "generic": 1, "fast": 2}[name])


def last_impl():
    from gvl_amd import _lib
    return {0: "none", 1: "generic", 2: "fast"}[_lib.lib().gvl_msda_last_impl()]


def fast_eligible(f):
    return (f["value"].dtype == np.float32 and f["value"].shape[3] == 64 and int(f["shapes"][:, 0].max()) == 1
            and f["loc"].shape[3] * f["loc"].shape[4] <= 16)


def tols(dtype):
    return (1e-10, 1e-10) if dtype == np.float64 else (1e-4, 1e-4)


def scale(a):
    return max(1.0, float(np.abs(a).max()))


@pytest.mark.parametrize("case", OP_CASES)
@pytest.mark.parametrize("pad", ["zeros", "border"])
@pytest.mark.parametrize("impl", ["generic", "fast"])
def test_op_matches_reference_golden(case, pad, impl, dev, MSDA):
    f = load(case)
    if impl == "fast" and not fast_eligible(f):
        pytest.skip("fast kernels cover fp32 / D=64 / temporal levels")
    atol, rtol = tols(f["value"].dtype)
    set_impl(impl)
    try:
        args = [t(f[k]).to(dev) for k in ("value", "shapes", "lsi", "loc", "aw")]
        out = MSDA.ms_deform_attn_forward(*args, 64, pad_mode=pad)
        assert last_impl() == impl
        assert maxerr(out, f[f"out_{pad}"]) <= atol * scale(f[f"out_{pad}"])
        gv, gl, gw = MSDA.ms_deform_attn_backward(*args, t(f["gout"]).to(dev), 64, pad_mode=pad)
        assert last_impl() == impl
        assert maxerr(gv, f[f"gvalue_{pad}"]) <= rtol * scale(f[f"gvalue_{pad}"])
        assert maxerr(gl, f[f"gloc_{pad}"]) <= rtol * scale(f[f"gloc_{pad}"])
        assert maxerr(gw, f[f"gaw_{pad}"]) <= rtol * scale(f[f"gaw_{pad}"])
    finally:
        set_impl("auto")


@pytest.mark.parametrize("case", OP_CASES)
def test_sample_matches_reference_golden(case, dev, MSDA):
    f = load(case)
    atol, _ = tols(f["value"].dtype)
    args = [t(f[k]).to(dev) for k in ("value", "shapes", "lsi", "loc")]
    s = MSDA.ms_deform_attn_sample(*args, pad_mode="border")
    assert maxerr(s, f["sample_border"]) <= atol * scale(f["sample_border"])


def make_inputs(B, T, M, D, Q, P, seed, dtype=np.float32, shapes2d=None, lo=-0.25, hi=1.25):
    rs = np.random.RandomState(seed)
    if shapes2d is None:
        lens = level_lengths(T)
        shapes = np.array([(1, x) for x in lens], np.int64)
    else:
        shapes = np.array(shapes2d, np.int64)
    L = len(shapes)
    sizes = shapes[:, 0] * shapes[:, 1]
    S = int(sizes.sum())
    lsi = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    value = rs.standard_normal((B, S, M, D)).astype(dtype)
    loc = rs.uniform(lo, hi, (B, Q, M, L, P, 2)).astype(dtype)
    if shapes[:, 0].max() == 1:
        loc[..., 1] = 0.5
    logits = rs.standard_normal((B, Q, M, L * P))
    aw = np.exp(logits - logits.max(-1, keepdims=True))
    aw = (aw / aw.sum(-1, keepdims=True)).reshape(B, Q, M, L, P).astype(dtype)
    gout = rs.standard_normal((B, Q, M * D)).astype(dtype)
    return value, shapes, lsi, loc, aw, gout


CFG = [
    # name, B, T, M, D, Q, P
    ("cfgA_dec", 16, 100, 8, 64, 300, 4),          # BASELINE.json config 1: decoder cross-attention
    ("cfgA_enc", 16, 100, 8, 64, 188, 4),          # encoder self-attention (Lq = S)
    ("ragged", 3, 37, 5, 64, 23, 3),               # Q not a multiple of 4, L*P = 12 < 16, odd sizes
    ("one_row_levels", 2, 4, 2, 64, 9, 4),         # levels 4,2,1,1 -> T_l == 1 edge case
    ("split_odd", 16, 37, 8, 64, 33, 4),           # B*M = 128: the level-split backward with an odd query count (17 + 16)
    ("split_q3", 16, 20, 8, 64, 3, 4),             # ... and with almost no queries (2 + 1)
]


def test_level_split_backward_equals_query_split(dev, MSDA):
    """k_bwd_t1d_split (two workgroups per slab own disjoint pyramid levels) against k_bwd_t1d_d64 + k_sum_partials on the
    same inputs: everything to summation order (round 6: the level-split kernel's own pass takes the dot products lane = sample,
    channel pairs in two chains -- the same products as the query-split kernel's lane = channels form, added in another order),
    and bit for bit against ITSELF on a second run (no float atomics anywhere: grad_loc / grad_attn are reproducible)."""
    value, shapes, lsi, loc, aw, gout = make_inputs(16, 100, 8, 64, 300, 4, seed=77)
    args = [t(x).to(dev) for x in (value, shapes, lsi, loc, aw)]
    res = {}
    from gvl_amd import _lib
    for split in ("1", "0"):
        os.environ["GVL_MSDA_BWD_SPLIT"] = split
        _lib.reload_env()                                    # (the library caches its switches)
        try:
            res[split] = MSDA.ms_deform_attn_backward(*args, t(gout).to(dev), 64)
        finally:
            os.environ.pop("GVL_MSDA_BWD_SPLIT", None)
            _lib.reload_env()
    (gv1, gl1, gw1), (gv0, gl0, gw0) = res["1"], res["0"]
    assert maxerr(gl1, gl0) <= 2e-6 * scale(gl0.cpu().numpy()) and maxerr(gw1, gw0) <= 2e-6 * scale(gw0.cpu().numpy())
    assert maxerr(gv1, gv0) <= 1e-5 * scale(gv0.cpu().numpy())
    _, gl2, gw2 = MSDA.ms_deform_attn_backward(*args, t(gout).to(dev), 64)
    assert torch.equal(gl1, gl2) and torch.equal(gw1, gw2)


@pytest.mark.parametrize("name,B,T,M,D,Q,P", CFG)
@pytest.mark.parametrize("pad", ["zeros", "border"])
@pytest.mark.parametrize("impl", ["generic", "fast"])
def test_op_matches_oracle_seeded(name, B, T, M, D, Q, P, pad, impl, dev, MSDA):
    from oracle import msda_oracle as O
    value, shapes, lsi, loc, aw, gout = make_inputs(B, T, M, D, Q, P, seed=zlib.crc32(name.encode()) % 1000)
    set_impl(impl)
    try:
        args = [t(x).to(dev) for x in (value, shapes, lsi, loc, aw)]
        out = MSDA.ms_deform_attn_forward(*args, 64, pad_mode=pad)
        assert last_impl() == impl
        ref = O.msda_forward(value, shapes, lsi, loc, aw, pad)
        assert maxerr(out, ref) <= 1e-4
        gv, gl, gw = MSDA.ms_deform_attn_backward(*args, t(gout).to(dev), 64, pad_mode=pad)
        rv, rl, rw = O.msda_backward(value, shapes, lsi, loc, aw, gout, pad)
        assert maxerr(gv, rv) <= 1e-4 * scale(rv)
        assert maxerr(gl, rl) <= 1e-4 * scale(rl)
        assert maxerr(gw, rw) <= 1e-4 * scale(rw)
    finally:
        set_impl("auto")


@pytest.mark.parametrize("D", [30, 32, 64, 71, 1025, 2048, 3096])       # pdvc/ops/test.py:85 channel list
def test_reference_test_geometry_f64(D, dev, MSDA):
    """pdvc/ops/test.py geometry: 2-D levels (6,4),(3,2), N=1, M=2, Lq=2, L=2, P=2, fp64, im2col_step=2; the
    analytic backward is checked against the oracle's (itself pinned to the reference's autograd)."""
    from oracle import msda_oracle as O
    value, shapes, lsi, loc, aw, gout = make_inputs(1, 0, 2, D, 2, 2, seed=3, dtype=np.float64,
                                                    shapes2d=[(6, 4), (3, 2)], lo=0.0, hi=1.0)
    value *= 0.01
    args = [t(x).to(dev) for x in (value, shapes, lsi, loc, aw)]
    for pad in ("zeros", "border"):
        out = MSDA.ms_deform_attn_forward(*args, 2, pad_mode=pad)
        assert maxerr(out, O.msda_forward(value, shapes, lsi, loc, aw, pad)) < 1e-12
        gv, gl, gw = MSDA.ms_deform_attn_backward(*args, t(gout).to(dev), 2, pad_mode=pad)
        rv, rl, rw = O.msda_backward(value, shapes, lsi, loc, aw, gout, pad)
        assert maxerr(gv, rv) < 1e-12 and maxerr(gl, rl) < 1e-10 and maxerr(gw, rw) < 1e-10


def test_autograd_function_gradcheck_f64(dev):
    """torch.autograd.gradcheck in fp64 as pdvc/ops/test.py:63-81 does (numerical vs analytic)."""
    from gvl_amd.ops.functions import MSDeformAttnFunction
    value, shapes, lsi, loc, aw, _ = make_inputs(1, 0, 2, 8, 2, 2, seed=5, dtype=np.float64,
                                                 shapes2d=[(6, 4), (3, 2)], lo=0.05, hi=0.95)
    v, l_, a = (t(x).to(dev).requires_grad_() for x in (value * 0.01, loc, aw))
    assert torch.autograd.gradcheck(MSDeformAttnFunction.apply, (v, t(shapes).to(dev), t(lsi).to(dev), l_, a, 2))


def test_full_size_properties(dev, MSDA):
    """BASELINE.json full size (B=16, T=100, Q=300): linearity in value and attention, fast == generic,
    run-to-run stability of the fast backward, grad_loc_y = -w * grad_w (H=1, zeros)."""
    value, shapes, lsi, loc, aw, gout = make_inputs(16, 100, 8, 64, 300, 4, seed=42)
    v, sh, ls, lc, a, g = (t(x).to(dev) for x in (value, shapes, lsi, loc, aw, gout))
    v2 = torch.randn_like(v)
    f = lambda vv, aa: MSDA.ms_deform_attn_forward(vv, sh, ls, lc, aa, 64)
    o1, o2, o12 = f(v, a), f(v2, a), f(v + 2 * v2, a)
    assert maxerr(o12, o1 + 2 * o2) < 2e-4
    assert maxerr(f(v, 3 * a), 3 * o1) < 2e-4
    set_impl("generic")
    og = f(v, a)
    gg = MSDA.ms_deform_attn_backward(v, sh, ls, lc, a, g, 64)
    set_impl("fast")
    of = f(v, a)
    gf1 = MSDA.ms_deform_attn_backward(v, sh, ls, lc, a, g, 64)
    gf2 = MSDA.ms_deform_attn_backward(v, sh, ls, lc, a, g, 64)
    set_impl("auto")
    assert maxerr(og, of) < 1e-4
    for x, y in zip(gg, gf1):
        assert maxerr(x, y) <= 1e-4 * scale(y.cpu().numpy())
    for x, y in zip(gf1[1:], gf2[1:]):
        assert torch.equal(x, y), "grad_loc / grad_attn involve no atomics and must be bitwise reproducible"
    assert maxerr(gf1[0], gf2[0]) <= 1e-5 * scale(gf1[0].cpu().numpy())     # no float atomics; equal-row entries are summed in counting-sort slot order, which varies
    assert maxerr(gf1[1][..., 1], -a * gf1[2]) < 1e-4 * scale(gf1[2].cpu().numpy())


def test_error_behaviour(dev, MSDA):
    value, shapes, lsi, loc, aw, gout = make_inputs(3, 10, 2, 64, 5, 4, seed=1)
    v, sh, ls, lc, a = (t(x).to(dev) for x in (value, shapes, lsi, loc, aw))
    with pytest.raises(RuntimeError, match="im2col_step"):           # ms_deform_attn_cuda.cu:50-52
        MSDA.ms_deform_attn_forward(v, sh, ls, lc, a, 2)
    with pytest.raises(RuntimeError, match="contiguous"):            # cu:28
        MSDA.ms_deform_attn_forward(v.transpose(1, 2), sh, ls, lc, a, 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):   # ms_deform_attn.h:38
        MSDA.ms_deform_attn_forward(v.cpu(), sh, ls, lc, a, 64)
    with pytest.raises(RuntimeError, match="dtype"):
        MSDA.ms_deform_attn_forward(v.half(), sh, ls, lc.half(), a.half(), 64)
    # empty query set: legal, returns an empty tensor; backward still zero-fills grad_value
    out = MSDA.ms_deform_attn_forward(v, sh, ls, lc[:, :0].contiguous(), a[:, :0].contiguous(), 64)
    assert out.shape == (3, 0, 2 * 64)
    gv, gl, gw = MSDA.ms_deform_attn_backward(v, sh, ls, lc[:, :0].contiguous(), a[:, :0].contiguous(),
                                              out.new_zeros(3, 0, 128), 64)
    assert float(gv.abs().max()) == 0.0 and gl.numel() == 0


def test_nan_and_far_locations_are_contained(dev, MSDA):
    """locations far outside / non-finite must not fault (rows are clamped inside the LDS slab) and, in zeros
    mode, far-outside samples contribute exactly 0."""
    value, shapes, lsi, loc, aw, gout = make_inputs(2, 20, 2, 64, 8, 4, seed=7)
    loc[0, 0, :, :, :, 0] = 1e9
    loc[0, 1, :, :, :, 0] = -1e9
    loc[1, 0, 0, 0, 0, 0] = np.nan
    v, sh, ls, lc, a, g = (t(x).to(dev) for x in (value, shapes, lsi, loc, aw, gout))
    for impl in ("generic", "fast"):
        set_impl(impl)
        out = MSDA.ms_deform_attn_forward(v, sh, ls, lc, a, 64)
        gv, gl, gw = MSDA.ms_deform_attn_backward(v, sh, ls, lc, a, g, 64)
        torch.cuda.synchronize()
        assert float(out[0, :2].abs().max()) == 0.0
        assert float(gl[0, :2].abs().max()) == 0.0 and float(gw[0, :2].abs().max()) == 0.0
        assert torch.isfinite(gv).all()
    set_impl("auto")


@pytest.mark.parametrize("pad", ["zeros", "border"])
def test_long_video_level0_in_global_matches_oracle(pad, dev, MSDA):
    """T = 512 (BASELINE config 5 / cfg L): S = 960 rows do not fit LDS; the temporal kernels keep level 0 in global
    memory and stage levels 1..3 (include/gvl_msda.h).  Forward + backward vs the CPU oracle; fast == generic."""
    from oracle import msda_oracle as O
    value, shapes, lsi, loc, aw, gout = make_inputs(2, 512, 8, 64, 77, 4, seed=512)
    args = [t(x).to(dev) for x in (value, shapes, lsi, loc, aw)]
    set_impl("fast")
    try:
        out = MSDA.ms_deform_attn_forward(*args, 64, pad_mode=pad)
        assert last_impl() == "fast"
        assert maxerr(out, O.msda_forward(value, shapes, lsi, loc, aw, pad)) <= 1e-4
        gv, gl, gw = MSDA.ms_deform_attn_backward(*args, t(gout).to(dev), 64, pad_mode=pad)
        assert last_impl() == "fast"
        rv, rl, rw = O.msda_backward(value, shapes, lsi, loc, aw, gout, pad)
        assert maxerr(gv, rv) <= 1e-4 * scale(rv)
        assert maxerr(gl, rl) <= 1e-4 * scale(rl)
        assert maxerr(gw, rw) <= 1e-4 * scale(rw)
    finally:
        set_impl("auto")


def test_randomised_shape_sweep_fast_vs_oracle(dev, MSDA):
    """Many small random temporal configurations (ragged Q, single query, single video/head, one-row levels, L*P < 16,
    workgroup-chunk boundaries) through the temporal kernels, against the CPU oracle, both paddings."""
    from oracle import msda_oracle as O
    rs = np.random.RandomState(2024)
    set_impl("fast")
    try:
        for trial in range(40):
            L = int(rs.randint(1, 5))
            P = int(rs.choice([p for p in (1, 2, 3, 4, 8, 16) if L * p <= 16]))
            lens = [int(rs.randint(1, 40)) for _ in range(L)]
            B, M, Q = int(rs.randint(1, 4)), int(rs.randint(1, 5)), int(rs.randint(1, 70))
            shapes = np.array([(1, x) for x in lens], np.int64)
            S = sum(lens)
            lsi = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
            value = rs.standard_normal((B, S, M, 64)).astype(np.float32)
            loc = rs.uniform(-0.3, 1.3, (B, Q, M, L, P, 2)).astype(np.float32)
            loc[..., 1] = 0.5
            aw = rs.rand(B, Q, M, L, P).astype(np.float32)
            aw /= aw.sum((-1, -2), keepdims=True)
            gout = rs.standard_normal((B, Q, M * 64)).astype(np.float32)
            pad = "zeros" if trial % 2 == 0 else "border"
            args = [t(x).to(dev) for x in (value, shapes, lsi, loc, aw)]
            out = MSDA.ms_deform_attn_forward(*args, 64, pad_mode=pad)
            assert last_impl() == "fast"
            tag = (trial, B, M, Q, lens, P, pad)
            assert maxerr(out, O.msda_forward(value, shapes, lsi, loc, aw, pad)) <= 1e-4, tag
            gv, gl, gw = MSDA.ms_deform_attn_backward(*args, t(gout).to(dev), 64, pad_mode=pad)
            rv, rl, rw = O.msda_backward(value, shapes, lsi, loc, aw, gout, pad)
            assert maxerr(gv, rv) <= 1e-4 * scale(rv), tag
            assert maxerr(gl, rl) <= 1e-4 * scale(rl), tag
            assert maxerr(gw, rw) <= 1e-4 * scale(rw), tag
    finally:
        set_impl("auto")


def test_general_y_coordinate_on_temporal_levels(dev, MSDA):
    """The op's API allows any y in [0,1] even for H = 1 levels (GVL always passes 0.5): the temporal kernels apply
    the vertical interpolation weight and its gradient like the reference kernel does (cuh:39-82 with H = 1)."""
    from oracle import msda_oracle as O
    value, shapes, lsi, loc, aw, gout = make_inputs(2, 30, 3, 64, 21, 4, seed=99)
    loc[..., 1] = np.random.RandomState(5).uniform(-0.6, 1.6, loc[..., 1].shape).astype(np.float32)
    args = [t(x).to(dev) for x in (value, shapes, lsi, loc, aw)]
    for impl in ("fast", "generic"):
        set_impl(impl)
        out = MSDA.ms_deform_attn_forward(*args, 64)
        assert maxerr(out, O.msda_forward(value, shapes, lsi, loc, aw, "zeros")) <= 1e-4
        gv, gl, gw = MSDA.ms_deform_attn_backward(*args, t(gout).to(dev), 64)
        rv, rl, rw = O.msda_backward(value, shapes, lsi, loc, aw, gout, "zeros")
        assert maxerr(gv, rv) <= 1e-4 * scale(rv) and maxerr(gl, rl) <= 1e-4 * scale(rl) and maxerr(gw, rw) <= 1e-4 * scale(rw)
    set_impl("auto")


@pytest.mark.parametrize("case", ["op_t1d_d64_f32", "op_2d_edges_f32", "op_test2d_d30_f64"])
def test_core_pytorch_named_entry_point(case, dev):
    """gvl_amd.ops.functions.ms_deform_attn_core_pytorch keeps the reference function's name / arguments / results
    (func.py:44-71, border padding, return_value) and its differentiability, on the GPU."""
    from gvl_amd.ops.functions import ms_deform_attn_core_pytorch
    f = load(case)
    atol, rtol = tols(f["value"].dtype)
    v, l_, a = (t(f[k]).to(dev).requires_grad_() for k in ("value", "loc", "aw"))
    out = ms_deform_attn_core_pytorch(v, t(f["shapes"]).to(dev), l_, a)
    assert maxerr(out, f["out_border"]) <= atol * scale(f["out_border"])
    out.backward(t(f["gout"]).to(dev))
    assert maxerr(v.grad, f["gvalue_border"]) <= rtol * scale(f["gvalue_border"])
    assert maxerr(l_.grad, f["gloc_border"]) <= rtol * scale(f["gloc_border"])
    assert maxerr(a.grad, f["gaw_border"]) <= rtol * scale(f["gaw_border"])
    s = ms_deform_attn_core_pytorch(t(f["value"]).to(dev), t(f["shapes"]).to(dev), t(f["loc"]).to(dev), None,
                                    return_value=True)
    assert maxerr(s, f["sample_border"]) <= atol * scale(f["sample_border"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ms_deform_attn_core_pytorch(t(f["value"]), t(f["shapes"]), t(f["loc"]), t(f["aw"]))
