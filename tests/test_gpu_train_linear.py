"""The dense layers of the TRAINING step on the hand-written kernels (gvl_amd/linear.py: train_linear; include/gvl_msda.h:
gvl_linear_f16x3_f32 on the planes of W and of W^T, gvl_wgrad_f16x3_f32, gvl_planes_refresh_f16) against float64 evaluations of
what autograd computes for nn.Linear (AddmmBackward: grad @ W, grad^T @ x, grad.sum(0)) -- every result must be at least as
close to float64 as the fp32 library GEMM's own result (factor 1.5 for the rounding of the comparison itself)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import gvl_amd  # noqa: E402,F401
from gvl_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402
from gvl_amd import layers as L  # noqa: E402
from gvl_amd import linear as GL  # noqa: E402
from gvl_amd.train_planes import TrainPlanes  # noqa: E402

DEV = "cuda:0"


def _err(a, ref):
    return ((a.double() - ref).abs().max() / ref.abs().max()).item()


@pytest.mark.parametrize("R,N,K", [(4800, 512, 512), (3008, 256, 512), (1000, 1536, 512), (4800, 512, 2048), (640, 64, 128),
                                   (2208, 8520, 512), (37, 128, 64)])
def test_wgrad_matches_float64(R, N, K):
    torch.manual_seed(R + N)
    # rows of very different magnitude (matched / unmatched queries), a few zero rows (padded positions)
    dy = torch.randn(R, N, device=DEV) * torch.exp(4 * torch.randn(R, 1, device=DEV)) * 1e-4
    dy[::7] = 0
    x = torch.randn(R, K, device=DEV) * 3
    am_dy, am_x = L.row_absmax(dy)[0], L.row_absmax(x)[0]
    gw, gb = MSDA.wgrad(dy, x, am_dy, am_x)
    ref, refb = dy.double().t() @ x.double(), dy.double().sum(0)
    lib = dy.t() @ x
    assert _err(gw, ref) <= max(1.5 * _err(lib, ref), 5e-7)               # (the split keeps 22 bits: 2^-21 on very short sums)
    assert _err(gb, refb) <= max(1.5 * _err(dy.sum(0), refb), 1e-6)        # (a sequential fp32 sum per thread, then 16 partials)
    # accumulate: on top of an existing gradient (AccumulateGrad folded in), and the strided (column-slice) form
    g2, b2 = torch.full_like(gw, 0.5), torch.full_like(gb, -0.25)
    MSDA.wgrad(dy, x, am_dy, am_x, grad_w=g2, grad_b=b2, accumulate=True)
    assert torch.allclose(g2, gw + 0.5, rtol=1e-5, atol=1e-6 * gw.abs().max().item())
    assert torch.allclose(b2, gb - 0.25, rtol=1e-5, atol=1e-6 * gb.abs().max().item())
    if N >= 128:
        half = N // 2 // 4 * 4
        gh, bh = MSDA.wgrad(dy[:, :half], x, am_dy, am_x)
        assert torch.allclose(gh, gw[:half], rtol=1e-5, atol=1e-6 * gw.abs().max().item())
        assert torch.allclose(bh, gb[:half], rtol=1e-5, atol=1e-6 * gb.abs().max().item())
    # deterministic: the split-K partials are added in a fixed order
    gw3, gb3 = MSDA.wgrad(dy, x, am_dy, am_x)
    assert torch.equal(gw3, gw) and torch.equal(gb3, gb)


def test_wgrad_bound_instead_of_maximum_and_tiny_values():
    torch.manual_seed(3)
    dy = torch.randn(1024, 128, device=DEV) * 1e-30           # far below fp16's range before scaling
    x = torch.randn(1024, 256, device=DEV) * 1e20
    one = lambda v: torch.full((1,), v, device=DEV)           # noqa: E731
    gw, _ = MSDA.wgrad(dy, x, one(8e-30), one(9e20))          # loose upper bounds instead of the exact maxima
    ref = dy.double().t() @ x.double()
    assert _err(gw, ref) < 2e-6
    z, _ = MSDA.wgrad(torch.zeros_like(dy), x, one(0.0), one(9e20))
    assert torch.count_nonzero(z) == 0 and torch.isfinite(z).all()


def test_planes_refresh_equals_per_matrix_split():
    """gvl_planes_refresh_f16: the planes of [W0; W1] and of its transpose reproduce the matrices to 22 bits, the bias is the
    concatenation, and a second refresh follows the parameters"""
    torch.manual_seed(1)
    w0, w1 = torch.randn(128, 512, device=DEV) * 0.05, torch.randn(128, 512, device=DEV) * 3e-4
    w2 = torch.randn(1536, 512, device=DEV)
    b0, b1 = torch.randn(128, device=DEV), torch.randn(128, device=DEV)
    tp = TrainPlanes(DEV)
    tp.register([w0, w1], [b0, b1])
    tp.register([w2], [None])
    for rnd in range(2):
        tp.refresh()
        for ws, bs in (([w0, w1], [b0, b1]), ([w2], None)):
            fwd, tr, bias = tp.lookup(ws)
            W = torch.cat(ws, 0)
            for planes, M in ((fwd, W), (tr, W.t())):
                hi, lo = planes.dense()
                rec = planes.scale[:, None].double() * (hi.double() + lo.double() / 2048)
                assert (rec - M.double()).abs().max().item() <= W.abs().max().item() * 2.0 ** -21
            if bs is not None:
                assert torch.equal(bias, torch.cat(bs))
        w0.mul_(1.7); w2.add_(0.3); b1.sub_(1.0)
        assert not tp.is_fresh()


@pytest.mark.parametrize("R,K,Ns,bias", [(4800, 512, (512,), True), (3008, 512, (128, 128), True), (1024, 2048, (512,), True),
                                          (4800, 512, (1536,), False)])
def test_train_linear_forward_and_gradients_match_float64(R, K, Ns, bias):
    torch.manual_seed(R + K)
    x = (torch.randn(16, R // 16, K, device=DEV) * 2).requires_grad_()
    ws = [torch.nn.Parameter(torch.randn(n, K, device=DEV) * 0.05) for n in Ns]
    bs = [torch.nn.Parameter(torch.randn(n, device=DEV)) if bias else None for n in Ns]
    assert GL.train_linear_eligible(x, ws, bs)
    g = torch.randn(16, R // 16, sum(Ns), device=DEV) * torch.exp(3 * torch.randn(16, R // 16, 1, device=DEV))

    def run(registered):
        for t in [x] + ws + [b for b in bs if b is not None]:
            t.grad = None
        prev = None
        if registered:
            tp = TrainPlanes(DEV)
            tp.register(ws, bs)
            tp.refresh()
            prev = GL.set_active_planes(tp)
        try:
            y = GL.train_linear(x, ws, bs)
            y.backward(g)
        finally:
            if registered:
                GL.set_active_planes(prev)
        return y.detach(), x.grad.clone(), [w.grad.clone() for w in ws], [b.grad.clone() if b is not None else None for b in bs]
    W64 = torch.cat([w.detach().double() for w in ws], 0)
    b64 = torch.cat([b.detach().double() for b in bs]) if bias else 0
    x64, g64 = x.detach().double().reshape(-1, K), g.double().reshape(-1, sum(Ns))
    ref_y, ref_gx, ref_gw, ref_gb = x64 @ W64.t() + b64, g64 @ W64, g64.t() @ x64, g64.sum(0)
    xf, gf, Wf = x.detach().reshape(-1, K), g.reshape(-1, sum(Ns)), torch.cat([w.detach() for w in ws], 0)
    lib = dict(y=_err(xf @ Wf.t() + (torch.cat([b.detach() for b in bs]) if bias else 0), ref_y), gx=_err(gf @ Wf, ref_gx),
               gw=_err(gf.t() @ xf, ref_gw))
    for registered in (False, True):
        y, gx, gws, gbs = run(registered)
        assert _err(y.reshape(-1, sum(Ns)), ref_y) <= 1.5 * lib["y"] + 1e-9
        assert _err(gx.reshape(-1, K), ref_gx) <= 1.5 * lib["gx"] + 1e-9
        assert _err(torch.cat(gws, 0), ref_gw) <= 1.5 * lib["gw"] + 1e-9
        if bias:
            assert _err(torch.cat(gbs), ref_gb) <= 2e-6


def test_linear_module_routes_training_products_to_the_hand_written_kernels():
    lin = GL.Linear(512, 512).to(DEV)
    x = torch.randn(16, 188, 512, device=DEV, requires_grad=True)
    MSDA.profile_enable(2)
    try:
        MSDA.profile_collect()
        lin(x).sum().backward()
        torch.cuda.synchronize()
        names = [t[0] for t in MSDA.profile_collect()]
    finally:
        MSDA.profile_enable(0)
    assert names.count("layer_gemm") == 2 and names.count("wgrad_f16x3") == 1, names      # forward, dx; dW + db
    ref = torch.nn.functional.linear(x.detach().double(), lin.weight.detach().double(), lin.bias.detach().double())
    assert _err(lin(x).detach(), ref) < 2e-6
    assert lin.weight.grad is not None and lin.bias.grad is not None and x.grad is not None
    GL.train_linear_enabled(False)
    try:
        assert not GL.train_linear_eligible(x, (lin.weight,), (lin.bias,))
    finally:
        GL.train_linear_enabled(True)


def test_split_k_product_for_a_long_contraction_and_the_vocabulary_planes():
    """gvl_linear_f16x3_splitk_f32 on the planes of W^T for W (8518, 512) -- N % 32 != 0: the transposed planes carry zeros for the
    contraction entries 8518 .. 8543 -- against float64; the forward planes of the same registration reproduce W"""
    torch.manual_seed(11)
    R, V, K = 2208, 8518, 512
    w = (torch.rand(V, K, device=DEV) - 0.5) * 0.2
    ld = (V + 31) // 32 * 32
    buf = torch.full((R, ld), float("nan"), device=DEV)
    g = buf[:, :V]
    g.copy_(torch.randn(R, V, device=DEV) * torch.exp(2 * torch.randn(R, 1, device=DEV)) * 1e-4)
    g[5] = 0                                                   # a masked-out caption row: bound 0
    buf[:, V:] = 0                                             # (what gvl_ce_rows_backward_f32 leaves in the padding)
    tp = TrainPlanes(DEV)
    tp.register([w], [None])
    tp.refresh()
    fwd, tr, _ = tp.lookup([w])
    hi, lo = fwd.dense()
    assert ((fwd.scale[:, None].double() * (hi.double() + lo.double() / 2048)) - w.double()).abs().max().item() <= 0.1 * 2.0 ** -21
    from gvl_amd.train_planes import Operand
    am = g.abs().amax(1)
    out = L.linear_splitk(buf, am, Operand(tr, K, ld, None))
    ref = g.double() @ w.double()
    assert _err(out, ref) <= max(1.5 * _err(g @ w, ref), 5e-7)
    assert torch.equal(out, L.linear_splitk(buf, am, Operand(tr, K, ld, None)))          # fixed summation order


def test_grouped_weight_gradients_equal_the_single_launches():
    """gvl_wgrad_group_f16x3_f32 against one gvl_wgrad_f16x3_f32 per problem: ragged rows, N not a multiple of the tile, a problem
    without bias, K = 1536 -- the same sums over other row ranges (<= 1e-6 of the largest element), and against float64"""
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    from gvl_amd import layers as L
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    probs = [(4800, 512, 512, True), (4800, 256, 512, True), (3008, 512, 512, False), (777, 96, 1536, True), (4800, 1024, 512, True),
             (130, 512, 64, True)]
    items, singles = [], []
    for R, N, K, bias in probs:
        dy = torch.randn(R, N, device=dev, generator=g) * 0.3
        x = torch.randn(R, K, device=dev, generator=g)
        am_dy, am_x = L.row_absmax(dy)[0], L.row_absmax(x)[0]
        items.append((dy, x, am_dy, am_x, torch.full((N, K), 7.0, device=dev), torch.full((N,), 7.0, device=dev) if bias else None))
        singles.append(MSDA.wgrad(dy, x, am_dy, am_x, want_bias=bias))
    MSDA.wgrad_group(items)
    for (dy, x, _, _, gw, gb), (sw, sb) in zip(items, singles):
        ref = dy.double().t() @ x.double()
        assert float((gw - sw).abs().max()) <= 1e-6 * float(sw.abs().max())
        assert float((gw.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
        if gb is not None:
            assert float((gb - sb).abs().max()) <= 1e-6 * float(sb.abs().max())
            assert float((gb.double() - dy.double().sum(0)).abs().max()) <= 2e-6 * float(dy.double().sum(0).abs().max())
        else:
            assert sb is None


def test_deferred_weight_gradients_of_a_train_step_equal_the_immediate_ones():
    """the layers' weight gradients taken in grouped launches when the backward pass ends (gvl_amd.linear.deferred_wgrads, what
    TrainStep does) against one launch pair per Linear inside each node's backward: every parameter's gradient equal to summation
    order, fewer weight-gradient launches, and a parameter used TWICE in the step (tied heads) still gets the sum of both."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import load, path_census
    from test_gpu_full_dims import build_anet, train_batch
    from gvl_amd import linear as GL
    kw = dict(transformer_dropout_prob=0.0, drop_prob=0.0)
    f, opt, model, crit = build_anet(True, **kw)
    dt = train_batch(f, load("pdvc_anet_full_train"))
    wd = crit.weight_dict

    def step(deferred):
        model.zero_grad(set_to_none=True)
        out, loss = model(dt, crit, None, "queries")
        final = sum(loss[k] * wd[k] for k in loss.keys() if k in wd)
        if deferred:
            with GL.deferred_wgrads():
                final.backward()
        else:
            final.backward()
        return {n: p_.grad.detach().clone() for n, p_ in model.named_parameters() if p_.grad is not None}
    ga, tags_a = path_census(lambda: step(False))
    gb, tags_b = path_census(lambda: step(True))
    assert ga.keys() == gb.keys()
    for n in ga:
        tol = 2e-6 * max(1e-6, float(ga[n].abs().max()))
        assert float((ga[n] - gb[n]).abs().max()) <= tol, (n, float((ga[n] - gb[n]).abs().max()), tol)
    wg = [k for k in tags_a if "wgrad" in k]
    assert wg and sum(tags_b[k] for k in wg) <= sum(tags_a[k] for k in wg) - 10, (tags_a, tags_b)


def test_a_captured_forward_of_a_stand_alone_linear_refreshes_its_planes_on_every_replay():
    """A Linear outside any model's TrainPlanes keeps private operand planes, refreshed when a parameter version moves.  A forward
    CAPTURED right after an eager forward at the same versions used to record no refresh: every replay then multiplied by the
    planes of the capture-time weights (found when the captioner's value_proj / ctx2att moved onto this path: data-parallel
    replicas drifted from the serial step).  Under capture the refresh is always recorded."""
    from gvl_amd.linear import Linear
    torch.manual_seed(3)
    lin = Linear(64, 128).to(DEV)
    x = torch.randn(768, 64, device=DEV, requires_grad=True)
    lin(x).sum().backward()                                   # eager: the private planes are built and fresh
    lin.zero_grad(set_to_none=True)
    static_x = x.detach().clone().requires_grad_()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            y = lin(static_x)
    torch.cuda.current_stream().wait_stream(side)
    with torch.no_grad():
        lin.weight.mul_(-2.0)                                 # (an optimizer step: in place, version bumped)
        lin.bias.add_(1.0)
    g.replay()
    torch.cuda.synchronize()
    want = torch.nn.functional.linear(static_x.detach().double(), lin.weight.detach().double(), lin.bias.detach().double())
    assert float((y.detach().double() - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("R,N,K", [(4416, 8518, 512), (1000, 512, 512), (96, 200, 64), (4416, 512, 2048)])
def test_wgrad_that_skips_the_stages_of_zero_rows_equals_the_full_pass(R, N, K):
    """gvl_wgrad_f16x3_live_f32: rows of dy whose bound is 0 are all zeros (the padded positions of a caption batch in the vocabulary
    layer's gradient); the product runs over the list of the other rows, 32 list entries per stage.  Same gradient as the full pass
    to rounding (the rows meet in other groups of 32) -- with most rows dead, with every row dead, with none"""
    torch.manual_seed(R + N)
    x = torch.randn(R, K, device=DEV)
    am_x = x.abs().amax(1)
    for frac_dead in (0.8, 1.0, 0.0):
        dy = torch.randn(R, N + (-N) % 4, device=DEV)[:, :N]
        dead = torch.rand(R, device=DEV) < frac_dead
        dead[: R // 3] = True if frac_dead > 0 else False                  # whole dead stages, not only scattered rows
        dy[dead] = 0.0
        am = dy.abs().amax(1)
        gw0, gb0 = MSDA.wgrad(dy, x, am, am_x)
        gw1, gb1 = MSDA.wgrad(dy, x, am, am_x, skip_zero_rows=True)
        ref = dy.double().t() @ x.double()
        scale = max(1e-30, float(ref.abs().max()))
        assert float((gw1.double() - ref).abs().max()) <= 3e-6 * max(scale, 1.0) and float((gw0.double() - ref).abs().max()) <= 3e-6 * max(scale, 1.0)
        assert float((gw1 - gw0).abs().max()) <= 2e-6 * max(scale, 1.0)
        assert float((gb1 - gb0).abs().max()) <= 1e-5 * max(1.0, float(gb0.abs().max()))
        if frac_dead == 1.0:
            assert float(gw1.abs().max()) == 0.0 and float(gb1.abs().max()) == 0.0
