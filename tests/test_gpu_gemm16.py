"""gvl_split_rows_f16 + gvl_gemm_f16x3_f32 (include/gvl_msda.h): the fp32 products of the captioner's token loop
(`self.logit`, `h2att` + recurrent gates, attention half of the LSTM input: pdvc/CaptioningHead/LSTM_DSA.py:121,165,247,
267-269) on the fp16 matrix cores.  The contract is fp32 accuracy: the error against an fp64 product must not exceed the
fp32 library GEMM's."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    return MSDA


def test_split_planes_reconstruct_the_input_to_22_bits():
    MSDA = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(257, 512, device=dev, generator=g) * torch.exp2(torch.randint(-30, 30, (257, 1), device=dev, generator=g).float())
    x[3] = 0.0                                                           # an all-zero row
    x[5, ::2] *= 1e-6                                                    # elements far below the row's maximum
    p = MSDA.split_rows(x)
    back = p.scale.double()[:, None] * (p.hi.double() + p.lo.double() / 2048.0)
    rowmax = x.abs().amax(1, keepdim=True).double()
    # per element: 2^-22 relative, or 2^-35 of the row maximum for elements that sit in fp16's subnormal range
    err = (back - x.double()).abs()
    assert bool((err <= 2.0 ** -22 * x.abs().double() + 2.0 ** -34 * rowmax).all())
    assert float(p.hi.abs().max()) <= 2.0 and bool(torch.isfinite(p.lo.float()).all())
    e = torch.log2(p.scale)
    assert bool((e == e.round()).all())                                  # scales are powers of two


@pytest.mark.parametrize("R,K,N", [(4800, 512, 8518), (4800, 512, 2560), (4800, 512, 2048), (37, 512, 8518),
                                   (130, 32, 70), (1, 64, 1), (300, 1024, 513)])
def test_product_error_is_within_the_fp32_library_gemm_s(R, K, N):
    MSDA = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(R + N)
    x = torch.randn(R, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    ref = x.double() @ w.double().t() + b.double()
    lib = torch.nn.functional.linear(x, w, b)
    mine = MSDA.gemm_f16x3(MSDA.split_rows(x), MSDA.split_rows(w), b)
    d_mine, d_lib = (mine.double() - ref), (lib.double() - ref)
    rms_mine, rms_lib = float(d_mine.pow(2).mean().sqrt()), float(d_lib.pow(2).mean().sqrt())
    ulp = 2.0 ** -23 * float(ref.abs().max())                            # the final rounding to fp32 alone (tiny cases)
    assert rms_mine <= 1.05 * rms_lib + ulp, (rms_mine, rms_lib)
    assert float(d_mine.abs().max()) <= 1.5 * float(d_lib.abs().max()) + ulp
    # without bias, into a wider output buffer (ldo > N)
    wide = torch.full((R, N + 5), 7.0, device=dev)
    MSDA.gemm_f16x3(MSDA.split_rows(x), MSDA.split_rows(w), None, out=wide[:, :N])
    assert float((wide[:, :N].double() - (ref - b.double())).abs().max()) <= 1.5 * float(d_lib.abs().max()) + ulp
    assert bool((wide[:, N:] == 7.0).all())


def test_product_of_operands_spread_over_many_orders_of_magnitude():
    MSDA = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(9)
    R, K, N = 512, 512, 640
    x = torch.randn(R, K, device=dev, generator=g) * torch.exp2(torch.randint(-20, 20, (R, K), device=dev, generator=g).float())
    w = torch.randn(N, K, device=dev, generator=g) * torch.exp2(torch.randint(-20, 20, (N, K), device=dev, generator=g).float())
    x[:, 0] = 1e30                                                        # far beyond fp16's range: the row scale absorbs it
    w[:, 0] = 1e-30
    ref = x.double() @ w.double().t()
    lib = x @ w.t()
    mine = MSDA.gemm_f16x3(MSDA.split_rows(x), MSDA.split_rows(w))
    assert bool(torch.isfinite(mine).all())
    bound = x.abs().double() @ w.abs().double().t()                      # error relative to sum |a||b|, like any GEMM's
    r_mine = float(((mine.double() - ref).abs() / bound).max())
    r_lib = float(((lib.double() - ref).abs() / bound).max())
    assert r_mine <= 2.0 * r_lib + 1e-9, (r_mine, r_lib)


def test_bad_arguments_are_refused():
    MSDA = _ops()
    dev = torch.device("cuda:0")
    with pytest.raises(RuntimeError):
        MSDA.split_rows(torch.randn(4, 30, device=dev))
    a, b = MSDA.split_rows(torch.randn(4, 64, device=dev)), MSDA.split_rows(torch.randn(6, 32, device=dev))
    with pytest.raises(RuntimeError):
        MSDA.gemm_f16x3(a, b)
