"""gvl_f16_products(1) (include/gvl_msda.h): ONE fp16 matrix-core product per fp32 product -- what inference under
torch.autocast runs the Linear layers on (gvl_amd/pdvc.py: autocast_inference_policy "f16").  The bar is the number format's:
both operands rounded to fp16 at their row scale, exact fp32 accumulation -> |error| <= 2^-10 sum |a||b| (+ the final rounding)
per output, whatever the kernel form; measured against bf16-rounded operands (what the reference's autocast multiplies) the
result is several times closer to the fp64 product.  Every form of the split-fp16 kernels is covered: the persistent
eight-wavefront kernels (store, fused argmax), the four-wavefront kernel (few rows, the LSTM cell epilogue) and the
inference layers' Linear kernel with its fused epilogues."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    return MSDA


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def test_switch_is_scoped_and_validated():
    MSDA = _ops()
    from gvl_amd import _lib
    L = _lib.lib()
    assert L.gvl_f16_products(0) == 3
    with MSDA.f16_products(1):
        assert L.gvl_f16_products(0) == 1
        with MSDA.f16_products(3):
            assert L.gvl_f16_products(0) == 3
        assert L.gvl_f16_products(0) == 1
    assert L.gvl_f16_products(0) == 3
    assert L.gvl_f16_products(2) < 0 and b"gvl_f16_products" in L.gvl_last_error()
    assert L.gvl_f16_products(0) == 3


@pytest.mark.parametrize("R,K,N", [(4800, 512, 8518), (4800, 512, 2560), (4800, 512, 2048), (37, 512, 8518), (130, 32, 70),
                                   (1, 64, 1), (300, 1024, 513), (2100, 96, 300)])
def test_single_product_error_is_the_fp16_rounding_of_the_operands(R, K, N):
    MSDA = _ops()
    x = _rand(R, K, seed=R + N) * torch.exp2(_rand(R, 1, seed=1, scale=4.0))          # rows of very different size
    w, b = _rand(N, K, seed=2, scale=0.05), _rand(N, seed=3)
    ref = x.double() @ w.double().t() + b.double()
    xp, wp = MSDA.split_rows(x), MSDA.split_rows(w)
    exact = MSDA.gemm_f16x3(xp, wp, b)
    with MSDA.f16_products(1):
        one = MSDA.gemm_f16x3(xp, wp, b)
    again = MSDA.gemm_f16x3(xp, wp, b)
    assert torch.equal(exact, again)                                           # the switch leaves nothing behind
    bound = 2.0 ** -10 * (x.abs().double() @ w.abs().double().t()) + 2.0 ** -22 * ref.abs() + 1e-30
    assert bool(((one.double() - ref).abs() <= bound).all())
    if R * N >= 10000 and K >= 96:
        rms_one = float((one.double() - ref).pow(2).mean().sqrt())
        rms_bf = float((x.bfloat16().double() @ w.bfloat16().double().t() + b.double() - ref).pow(2).mean().sqrt())
        rms_exact = float((exact.double() - ref).pow(2).mean().sqrt())
        assert rms_exact * 50 < rms_one < rms_bf / 4, (rms_exact, rms_one, rms_bf)   # 11-bit operands: 8x closer than bf16's 8


@pytest.mark.parametrize("R,V", [(4800, 8518), (1024, 2000), (40, 300)])
def test_fused_argmax_single_product(R, V):
    """the fused vocabulary form agrees with the argmax / log-softmax of the logits the single-product store form writes"""
    MSDA = _ops()
    K = 512
    x, w, b = _rand(R, K, seed=4), _rand(V, K, seed=5, scale=0.05), _rand(V, seed=6)
    xp, wp = MSDA.split_rows(x), MSDA.split_rows(w)
    with MSDA.f16_products(1):
        logits = MSDA.gemm_f16x3(xp, wp, b)
        tok, lp = MSDA.row_argmax_lse_partials(MSDA.gemm_f16x3_argmax(xp, wp, b))
    ls = torch.log_softmax(logits.double(), 1)
    top2 = logits.topk(2, 1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4                                   # (the two forms sum in different orders)
    assert torch.equal(tok[clear], logits.argmax(1)[clear])
    assert float((lp.double() - ls.gather(1, tok[:, None]).squeeze(1)).abs().max()) <= 2e-5


@pytest.mark.parametrize("n,form", [(4800, None), (4800, "8"), (300, None)])
def test_lstm_cell_epilogue_single_product(n, form, monkeypatch):
    MSDA = _ops()
    if form:
        monkeypatch.setenv("GVL_LSTM_GEMM_FORM", form)
    K = H = 512
    att, wg = _rand(n, K, seed=7), _rand(4 * H, K, seed=8, scale=0.05)
    gates_h, gates_c = _rand(n, 4 * H, seed=9), _rand(n, 4 * H, seed=10)
    emb = _rand(101, 4 * H, seed=11)
    it = torch.randint(0, 101, (n,), device=DEV)
    c = _rand(n, H, seed=12)
    perm = MSDA.gate_permutation(H, torch.device(DEV))
    ap, wpp = MSDA.split_rows(att), MSDA.split_rows(wg[perm].contiguous())
    ghp, gcp, embp = gates_h[:, perm].contiguous(), gates_c[:, perm].contiguous(), emb[:, perm].contiguous()
    gates = att.double() @ wg.double().t() + gates_h.double() + gates_c.double() + emb.double()[it]
    i_, f_, g_, o_ = gates.chunk(4, 1)
    c_ref = torch.sigmoid(f_) * c.double() + torch.sigmoid(i_) * torch.tanh(g_)
    h_ref = torch.sigmoid(o_) * torch.tanh(c_ref)
    h3, c3 = MSDA.gemm_f16x3_lstm(ap, wpp, ghp, gcp, embp, it, c)
    with MSDA.f16_products(1):
        h1, c1 = MSDA.gemm_f16x3_lstm(ap, wpp, ghp, gcp, embp, it, c)
    e3 = float((c3.double() - c_ref).abs().max())
    e1 = float((c1.double() - c_ref).abs().max())
    assert e3 <= 1e-5 and e3 * 20 < e1 <= 5e-3, (e3, e1)
    assert float((h1.double() - h_ref).abs().max()) <= 5e-3
    # h' leaves as planes for the next products of the step: they reconstruct h' itself
    p = h1._gvl_planes
    hi, lo = p.dense()
    back = p.scale[:, None] * (hi.float() + lo.float() / 2048.0)
    assert float((back - h1).abs().max()) <= 2.0 ** -20


def test_linear_kernel_single_product_with_every_epilogue():
    """gvl_linear_f16x3_f32 in single-product mode: segments, addend, masked rows, ReLU + residual, row maxima -- against
    the same launch in exact mode (the epilogues are shared, only the product differs)"""
    from gvl_amd import layers as L
    MSDA = _ops()
    R, K = 1507, 512
    x, pos = _rand(R, K, seed=5), _rand(R, K, seed=6, scale=2.0)
    blocks = [(_rand(512, K, seed=7, scale=0.05), _rand(512, seed=8)), (_rand(256, K, seed=9, scale=0.05), _rand(256, seed=10)),
              (_rand(100, K, seed=11, scale=0.05), _rand(100, seed=12))]
    res = _rand(R, 128, seed=13)
    mask = (torch.arange(R, device=DEV) % 7 == 3)
    am, amp = L.row_absmax(x, pos)
    W = L.Weights(blocks)

    def run():
        o0, o1, o2 = (torch.full((R, n), float("nan"), device=DEV) for n in (512, 256, 128))
        am2 = torch.zeros(R, device=DEV)
        L.linear(x, W, [L.seg(0, o0, am, rowmask=mask), L.seg(512, o1, amp, addend=True),
                        L.seg(768, o2, am, resid=res, amax_out=am2, relu=True, width=100)], a2=pos)
        return o0, o1, o2[:, :100], am2
    exact = run()
    with MSDA.f16_products(1):
        one = run()
    for a_, b_, scale in zip(one[:3], exact[:3], (1.0, 3.0, 1.0)):
        d = float((a_ - b_).abs().max())
        assert 0 < d <= 4e-3 * scale, d
    assert bool((one[0][mask] == 0).all())
    assert torch.equal(one[3], one[2].abs().amax(1))                          # row maxima of what was actually stored
    # wide tile (128 x 128) and the long-K shape of the FFN's second Linear
    for (R2, K2, N2) in ((4800, 512, 2048), (3008, 2048, 512)):
        x2, w2, b2 = _rand(R2, K2, seed=20), _rand(N2, K2, seed=21, scale=K2 ** -0.5), _rand(N2, seed=22)
        am2, _ = L.row_absmax(x2)
        W2 = L.Weights([(w2, b2)])
        want = x2.double() @ w2.double().t() + b2.double()
        out = torch.empty(R2, N2, device=DEV)
        with MSDA.f16_products(1):
            L.linear(x2, W2, [L.seg(0, out, am2)])
        bound = 2.0 ** -10 * (x2.abs().double() @ w2.abs().double().t()) + 2.0 ** -22 * want.abs()
        assert bool(((out.double() - want).abs() <= bound).all())
