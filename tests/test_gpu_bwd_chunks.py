"""The backward's looping form (k_bwd_t1d_d64<..., LOOP = true>): when the queries of a (b,m) slab do not fit one LDS
carve-up, the slab's workgroups walk several query chunks and accumulate the grad_value rows in place.  It is selected
when there are more chunks than workgroups per slab (B*M >= 256 -> one workgroup per slab), which the other op tests
(small B) never reach.  Checked against the CPU oracle and the generic kernel, both paddings, for
  * a long video (T = 512: level 0 in global memory, chunks forced by the LDS budget),
  * the cfg A shape with the chunk count forced through GVL_MSDA_BWD_CHUNKS,
  * the fused module entry point against the autograd composition of the unfused op."""
import numpy as np
import pytest
import torch

from helpers import t, maxerr
from test_gpu_op import make_inputs, set_impl, last_impl, scale

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def MSDA():
    from gvl_amd import MultiScaleDeformableAttention as m, _lib
    _lib.lib()
    return m


def _check(MSDA, dev, B, T, Q, pad, seed, oracle_rows=4, expect_kernel=None):
    from oracle import msda_oracle as O
    value, shapes, lsi, loc, aw, gout = make_inputs(B, T, 8, 64, Q, 4, seed=seed)
    args = [t(x).to(dev) for x in (value, shapes, lsi, loc, aw)]
    g = t(gout).to(dev)
    try:
        set_impl("fast")
        gv, gl, gw = MSDA.ms_deform_attn_backward(*args, g, 64, pad_mode=pad)
        assert last_impl() == "fast"
        if expect_kernel:
            from gvl_amd import _lib
            got = _lib.lib().gvl_msda_last_kernel().decode()
            assert got == expect_kernel or (isinstance(expect_kernel, tuple) and got in expect_kernel), got
        set_impl("generic")
        rv, rl, rw = MSDA.ms_deform_attn_backward(*args, g, 64, pad_mode=pad)
    finally:
        set_impl("auto")
    sv = scale(rv.cpu().numpy())
    assert maxerr(gv, rv.cpu().numpy()) <= 1e-4 * sv
    assert maxerr(gl, rl.cpu().numpy()) <= 1e-4 * scale(rl.cpu().numpy())
    assert maxerr(gw, rw.cpu().numpy()) <= 1e-4 * scale(rw.cpu().numpy())
    # the first videos once more against the CPU oracle (the generic kernel is pinned to it elsewhere)
    nb = min(B, oracle_rows)
    ov, ol, ow = O.msda_backward(value[:nb], shapes, lsi, loc[:nb], aw[:nb], gout[:nb], pad)
    assert maxerr(gv[:nb], ov) <= 1e-4 * scale(ov)
    assert maxerr(gl[:nb], ol) <= 1e-4 * scale(ol)
    assert maxerr(gw[:nb], ow) <= 1e-4 * scale(ow)


@pytest.mark.parametrize("pad", ["zeros", "border"])
def test_long_video_chunks_accumulate_in_place(pad, dev, MSDA, monkeypatch):
    # B*M = 256 slabs -> one workgroup each; 400 queries beside a 449-row slab need >= 3 chunks
    monkeypatch.setenv("GVL_MSDA_BWD_OWN", "0")
    _check(MSDA, dev, B=32, T=512, Q=400, pad=pad, seed=41, expect_kernel="k_bwd_t1d_d64<loop>")


@pytest.mark.parametrize("chunks", [2, 3, 5])
def test_forced_chunk_counts_at_the_training_shape(chunks, dev, MSDA, monkeypatch):
    monkeypatch.setenv("GVL_MSDA_BWD_OWN", "0")
    monkeypatch.setenv("GVL_MSDA_BWD_CHUNKS", str(chunks))
    _check(MSDA, dev, B=32, T=100, Q=300, pad="zeros", seed=7 + chunks)


def test_ragged_last_chunk_and_two_workgroups_per_slab(dev, MSDA, monkeypatch):
    # B*M = 128 -> two workgroups per slab; 6 chunks of 50 queries over Q = 277: three chunks each, the last one short
    monkeypatch.setenv("GVL_MSDA_BWD_CHUNKS", "6")
    _check(MSDA, dev, B=16, T=100, Q=277, pad="border", seed=99)


@pytest.mark.parametrize("T,Q", [(50, 300), (100, 400), (24, 512)])
def test_level_split_with_more_own_queries_than_slab_rows(T, Q, dev, MSDA, monkeypatch):
    """k_bwd_t1d_split stages the workgroup's own grad_out rows over the dead value slab; with
    (Q + 1) / 2 > S + 1 (short videos, many queries) those rows used to run into the foreign queries' rows.  The first
    LDS region is now sized for both uses; checked against the generic kernel / oracle and against the query-split form."""
    _check(MSDA, dev, B=16, T=T, Q=Q, pad="zeros", seed=300 + T)
    value, shapes, lsi, loc, aw, gout = make_inputs(16, T, 8, 64, Q, 4, seed=77)
    args = [t(x).to(dev) for x in (value, shapes, lsi, loc, aw)]
    g = t(gout).to(dev)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("GVL_MSDA_BWD_SPLIT", flag)
        res[flag] = [x.cpu().numpy() for x in MSDA.ms_deform_attn_backward(*args, g, 64)]
    for a, b in zip(res["1"], res["0"]):
        assert maxerr(a, b) <= 2e-5 * scale(b)


@pytest.mark.parametrize("ref_dim", [1, 2])
def test_fused_entry_point_with_chunks_matches_autograd_composition(ref_dim, dev, MSDA, monkeypatch):
    from gvl_amd.ops.functions.ms_deform_attn_func import MSDeformAttnFunction
    monkeypatch.setenv("GVL_MSDA_BWD_OWN", "0")                 # keep this test on the chunked form it was written for
    monkeypatch.setenv("GVL_MSDA_BWD_CHUNKS", "4")
    B, M, D, L, P, Q = 32, 8, 64, 4, 4, 120
    lens = [100, 50, 25, 13]
    S = sum(lens)
    gen = torch.Generator(device="cpu").manual_seed(5)
    value = torch.randn(B, S, M, D, generator=gen).to(dev)
    proj = (torch.randn(B, Q, 2 * M * L * P, generator=gen) * 0.7).to(dev)
    ref = torch.rand(B, Q, L, ref_dim, generator=gen).to(dev)
    if ref_dim == 2:
        ref[..., 1] = ref[..., 1] * 0.3 + 0.05
    gout = torch.randn(B, Q, M * D, generator=gen).to(dev)
    shapes = torch.tensor([(1, x) for x in lens], dtype=torch.long, device=dev)
    lsi = torch.tensor(np.concatenate([[0], np.cumsum(lens)[:-1]]), dtype=torch.long, device=dev)
    gv, gp, gr = MSDA.msda1d_fused_backward(value, shapes, lsi, proj, ref, gout, L, P, need_ref_grad=True)

    v2, p2, r2 = value.clone().requires_grad_(), proj.clone().requires_grad_(), ref.clone().requires_grad_()
    off = p2[..., :M * L * P].view(B, Q, M, L, P)
    w = torch.softmax(p2[..., M * L * P:].view(B, Q, M, L * P), -1).view(B, Q, M, L, P)
    T_l = torch.tensor(lens, dtype=torch.float32, device=dev)
    if ref_dim == 1:                                     # ms_deform_attn.py:103-106
        x = r2[:, :, None, :, None, 0] + off / T_l[None, None, None, :, None]
    else:                                                # :107-109
        x = r2[:, :, None, :, None, 0] + off / P * r2[:, :, None, :, None, 1] * 0.5
    loc = torch.stack([x, torch.full_like(x, 0.5)], -1)
    out = MSDeformAttnFunction.apply(v2, shapes, lsi, loc, w, 64)
    out.backward(gout)
    assert maxerr(gv, v2.grad.cpu().numpy()) <= 1e-4 * scale(v2.grad.cpu().numpy())
    assert maxerr(gp, p2.grad.cpu().numpy()) <= 1e-4 * scale(p2.grad.cpu().numpy())
    assert maxerr(gr, r2.grad.cpu().numpy()) <= 1e-4 * scale(r2.grad.cpu().numpy())


# ---- row-ownership form (k_bwd_t1d_own, round 4): two workgroups per slab, rows owned across query chunks -------------
def last_kernel():
    from gvl_amd import _lib
    return _lib.lib().gvl_msda_last_kernel().decode()


@pytest.mark.parametrize("T,Q", [(512, 300), (512, 960), (512, 777), (200, 375), (300, 450)])
@pytest.mark.parametrize("pad", ["zeros", "border"])
def test_row_ownership_backward_matches_generic_and_oracle(T, Q, pad, dev, MSDA, monkeypatch):
    """(GVL_MSDA_BWD_OWN_ALWAYS: the form is selected wherever it is ELIGIBLE -- by default short slabs that fit one carve-up stay
    on the chunked form, which is faster there; the kernel must be right on them all the same.)  B*M = 128 (two workgroups per slab) with more queries than one LDS carve-up holds: every grad_value row is owned
    by one wavefront across all query chunks and written once (no workspace, no k_sum_partials).  T = 512: level 0 read
    from global memory, 1-3 chunks; T = 200 / 300: whole slab in LDS, 1-2 chunks, ragged last chunk / odd halves."""
    from gvl_amd import _lib
    monkeypatch.setenv("GVL_MSDA_BWD_OWN_ALWAYS", "1")
    _check(MSDA, dev, B=16, T=T, Q=Q, pad=pad, seed=1000 + T + Q, oracle_rows=2, expect_kernel="k_bwd_t1d_own")
    from helpers import level_lengths
    lens = level_lengths(T)
    arr = (__import__("ctypes").c_int64 * 8)(*[v for x in lens for v in (1, x)])
    assert _lib.lib().gvl_msda_backward_workspace_bytes(16, sum(lens), 8, 64, 4, Q, 4, 4, arr) == 0
    assert _lib.lib().gvl_msda_backward_workspace_bytes(16, sum(lens), 8, 64, 4, Q, 4, 2, arr) == 0


def test_row_ownership_forced_small_chunks(dev, MSDA, monkeypatch):
    # 64 queries per chunk: 15 chunks over Q = 960, and a video count that gives B*M = 136 (not a multiple of 128)
    monkeypatch.setenv("GVL_MSDA_BWD_OWN_QC", "64")
    _check(MSDA, dev, B=17, T=512, Q=960, pad="zeros", seed=4242, oracle_rows=1, expect_kernel="k_bwd_t1d_own")


def test_row_ownership_equals_the_chunked_form(dev, MSDA, monkeypatch):
    """same inputs through k_bwd_t1d_own and through the query-chunked k_bwd_t1d_d64<loop> + k_sum_partials it replaces:
    grad_loc / grad_attn bit-equal (same arithmetic), grad_value equal up to the summation order of a row's entries"""
    value, shapes, lsi, loc, aw, gout = make_inputs(16, 512, 8, 64, 400, 4, seed=31)
    args = [t(x).to(dev) for x in (value, shapes, lsi, loc, aw)]
    g = t(gout).to(dev)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("GVL_MSDA_BWD_OWN", flag)
        res[flag] = [x.cpu().numpy() for x in MSDA.ms_deform_attn_backward(*args, g, 64)]
        assert last_kernel() == ("k_bwd_t1d_own" if flag == "1" else "k_bwd_t1d_d64<loop>")
    assert maxerr(res["1"][0], res["0"][0]) <= 2e-5 * scale(res["0"][0])
    assert np.array_equal(res["1"][1], res["0"][1]) and np.array_equal(res["1"][2], res["0"][2])


@pytest.mark.parametrize("ref_dim", [1, 2])
@pytest.mark.parametrize("T,Q", [(512, 300), (512, 960)])
def test_row_ownership_fused_matches_autograd_composition(T, Q, ref_dim, dev, MSDA):
    from gvl_amd.ops.functions.ms_deform_attn_func import MSDeformAttnFunction
    from helpers import level_lengths
    B, M, D, L, P = 16, 8, 64, 4, 4
    lens = level_lengths(T)
    S = sum(lens)
    gen = torch.Generator(device="cpu").manual_seed(50 + ref_dim + Q)
    value = torch.randn(B, S, M, D, generator=gen).to(dev)
    proj = (torch.randn(B, Q, 2 * M * L * P, generator=gen) * 0.7).to(dev)
    ref = torch.rand(B, Q, L, ref_dim, generator=gen).to(dev)
    if ref_dim == 2:
        ref[..., 1] = ref[..., 1] * 0.3 + 0.05
    gout = torch.randn(B, Q, M * D, generator=gen).to(dev)
    shapes = torch.tensor([(1, x) for x in lens], dtype=torch.long, device=dev)
    lsi = torch.tensor(np.concatenate([[0], np.cumsum(lens)[:-1]]), dtype=torch.long, device=dev)
    gv, gp, gr = MSDA.msda1d_fused_backward(value, shapes, lsi, proj, ref, gout, L, P, need_ref_grad=True)
    assert last_kernel() == "k_bwd_t1d_own"
    v2, p2, r2 = value.clone().requires_grad_(), proj.clone().requires_grad_(), ref.clone().requires_grad_()
    off = p2[..., :M * L * P].view(B, Q, M, L, P)
    w = torch.softmax(p2[..., M * L * P:].view(B, Q, M, L * P), -1).view(B, Q, M, L, P)
    T_l = torch.tensor(lens, dtype=torch.float32, device=dev)
    if ref_dim == 1:                                     # ms_deform_attn.py:103-106
        x = r2[:, :, None, :, None, 0] + off / T_l[None, None, None, :, None]
    else:                                                # :107-109
        x = r2[:, :, None, :, None, 0] + off / P * r2[:, :, None, :, None, 1] * 0.5
    loc = torch.stack([x, torch.full_like(x, 0.5)], -1)
    try:
        set_impl("generic")                              # the reference side: generic kernels (pinned to the oracle)
        out = MSDeformAttnFunction.apply(v2, shapes, lsi, loc, w, 64)
        out.backward(gout)
    finally:
        set_impl("auto")
    assert maxerr(gv, v2.grad.cpu().numpy()) <= 1e-4 * scale(v2.grad.cpu().numpy())
    assert maxerr(gp, p2.grad.cpu().numpy()) <= 1e-4 * scale(p2.grad.cpu().numpy())
    assert maxerr(gr, r2.grad.cpu().numpy()) <= 1e-4 * scale(r2.grad.cpu().numpy())


@pytest.mark.parametrize("T,Q", [(512, 300), (512, 960)])
def test_row_ownership_bf16_storage_equals_rounded_fp32(T, Q, dev, MSDA):
    """bf16 storage through the same kernel: rows leave in bf16 directly (no fp32 partial slabs, no rounding pass);
    grad_loc / grad_attn bit-equal to the fp32 kernel's, grad_value = round(fp32 result) up to entry order"""
    BF = torch.bfloat16
    value, shapes, lsi, loc, aw, gout = make_inputs(16, T, 8, 64, Q, 4, seed=T + Q + 5)
    v_bf, g_bf = t(value).to(dev).to(BF), t(gout).to(dev).to(BF)
    sh, ls, lc, w = (t(x).to(dev) for x in (shapes, lsi, loc, aw))
    gv, gl, gw = MSDA.ms_deform_attn_backward(v_bf, sh, ls, lc, w, g_bf, 64)
    assert last_kernel() == "k_bwd_t1d_own" and gv.dtype == BF
    gv32, gl32, gw32 = MSDA.ms_deform_attn_backward(v_bf.float(), sh, ls, lc, w, g_bf.float(), 64)
    assert last_kernel() == "k_bwd_t1d_own"
    # grad_value: the same fp32 gather, but the order of one row's entries follows the integer LDS atomics:
    # equal before rounding up to fp32 summation order, i.e. within one bf16 ulp, on a tiny fraction of the elements
    assert maxerr(gv.float(), gv32) <= 2.0 ** -8 * scale(gv32.cpu().numpy())
    assert float((gv != gv32.to(BF)).float().mean()) < 2e-3
    assert torch.equal(gl, gl32) and torch.equal(gw, gw32)


@pytest.mark.parametrize("T,Q,pad", [(24, 1500, "zeros"), (131, 333, "border"), (257, 64, "zeros"), (600, 130, "zeros"),
                                     (411, 901, "border"), (512, 2, "zeros")])
def test_row_ownership_odd_shapes(T, Q, pad, dev, MSDA, monkeypatch):
    """short videos with very many queries (tiny levels, many chunks), odd level lengths, a level length just past a power
    of two, the longest video the register accumulators hold (T = 600 -> 675 owned rows... of which {0,3} = 600 + 75 > 640
    falls back to the chunked kernel: asserted), two queries"""
    from gvl_amd import _lib
    from helpers import level_lengths
    monkeypatch.setenv("GVL_MSDA_BWD_OWN_ALWAYS", "1")
    lens = level_lengths(T)
    own = max(lens[0] + lens[3], lens[1] + lens[2]) <= 640
    _check(MSDA, dev, B=16, T=T, Q=Q, pad=pad, seed=7000 + T + Q, oracle_rows=1,
           expect_kernel=("k_bwd_t1d_own", "k_bwd_t1d_split") if own else None)      # (split: the queries fit one carve-up)
    if not own:
        assert _lib.lib().gvl_msda_last_kernel().decode() != "k_bwd_t1d_own"


@pytest.mark.parametrize("B,T,Q", [(32, 512, 700), (40, 512, 960), (24, 512, 300), (8, 512, 500), (12, 512, 960)])
def test_row_ownership_beyond_one_round_of_workgroups(B, T, Q, dev, MSDA):
    """B*M > 128: 2 B*M workgroups run in rounds; the two workgroups of a slab take adjacent positions on their XCD; B*M = 64,
    96: fewer workgroups than CUs (still ahead of the chunked form's partial slabs).  B = 32
    with few chunks stays on the chunked form (test_long_video_chunks_accumulate_in_place: 3 chunks)."""
    _check(MSDA, dev, B=B, T=T, Q=Q, pad="zeros", seed=9000 + B + Q, oracle_rows=1, expect_kernel="k_bwd_t1d_own")


def test_one_workgroup_per_slab_single_chunk_keeps_the_plain_form(dev, MSDA):
    # B*M = 256 at cfg A: everything fits one carve-up -> k_bwd_t1d_d64 (no chunks, no pair form)
    _check(MSDA, dev, B=32, T=100, Q=300, pad="zeros", seed=123, oracle_rows=1, expect_kernel="k_bwd_t1d_d64")


def test_short_slabs_stay_on_the_faster_forms(dev, MSDA):
    """the default selection: a slab that fits one LDS carve-up per workgroup is not handed to the pair form (B = 8, T = 100:
    21 us chunked + partial sum against 35 us; B = 24, T = 200: 53 against 64 us), long slabs are"""
    for B, T, Q, want in ((8, 100, 300, "k_bwd_t1d_d64"), (24, 200, 300, "k_bwd_t1d_d64"), (16, 100, 300, "k_bwd_t1d_split"),
                          (16, 512, 300, "k_bwd_t1d_own"), (12, 512, 960, "k_bwd_t1d_own")):
        _check(MSDA, dev, B=B, T=T, Q=Q, pad="zeros", seed=B + T + Q, oracle_rows=1, expect_kernel=want)
