#!/usr/bin/env python3
"""Dev tool (round 4): the split-fp16 products with 3 fp16 products per fp32 product (exact) against 1 (gvl_f16_products(1):
inference under autocast) -- error against fp64 beside bf16-operand rounding, and back-to-back launch times."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import _lib                                                    # noqa: E402
from gvl_amd import MultiScaleDeformableAttention as MSDA                  # noqa: E402
from gvl_amd import layers                                                  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (R, K, N, what) in ((4800, 512, 2560, "h product"), (4800, 512, 8518, "vocabulary (store)"), (4800, 2048, 512, "ffn2-like")):
    x = torch.randn(R, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    ref = x.double() @ w.double().t() + b.double()
    xp, wp = MSDA.split_rows(x), MSDA.split_rows(w)
    res = {}
    for n in (3, 1):
        L.gvl_f16_products(n)
        out = MSDA.gemm_f16x3(xp, wp, b)
        err = (out.double() - ref).abs()
        res[n] = (err.max().item(), err.pow(2).mean().sqrt().item(), timed(lambda: MSDA.gemm_f16x3(xp, wp, b)))
    L.gvl_f16_products(3)
    bf = (x.bfloat16().double() @ w.bfloat16().double().t() + b.double())
    e_bf = (bf - ref).abs()
    lib = timed(lambda: torch.addmm(b, x, w.t()))
    xb, wb, bb = x.bfloat16(), w.bfloat16(), b.bfloat16()
    libb = timed(lambda: torch.addmm(bb, xb, wb.t()))
    print(f"{what:20s} {R}x{K}x{N}: x3 max {res[3][0]:.2e} rms {res[3][1]:.2e} {res[3][2]:6.1f} us | x1 max {res[1][0]:.2e} rms "
          f"{res[1][1]:.2e} {res[1][2]:6.1f} us | bf16-rounded operands max {e_bf.max().item():.2e} rms "
          f"{e_bf.pow(2).mean().sqrt().item():.2e} | library fp32 {lib:6.1f} us, bf16 {libb:6.1f} us")

# the fused argmax form
R, K, V = 4800, 512, 8518
x = torch.randn(R, K, device=dev, generator=g)
w = torch.randn(V, K, device=dev, generator=g) * 0.05
b = torch.randn(V, device=dev, generator=g)
xp, wp = MSDA.split_rows(x), MSDA.split_rows(w)
ref = (x.double() @ w.double().t() + b.double())
tok_ref = ref.argmax(1)
for n in (3, 1):
    L.gvl_f16_products(n)
    tok, lp = MSDA.row_argmax_lse_partials(MSDA.gemm_f16x3_argmax(xp, wp, b))
    t = timed(lambda: MSDA.gemm_f16x3_argmax(xp, wp, b))
    lp_ref = torch.log_softmax(ref, 1).gather(1, tok[:, None]).squeeze(1)
    print(f"argmax form x{n}: {t:6.1f} us, tokens equal {(tok == tok_ref).float().mean().item():.4f}, "
          f"|logp - ref| max {(lp.double() - lp_ref).abs().max().item():.2e}")
L.gvl_f16_products(3)

# gate product with the cell in its epilogue
n, K, H = 4800, 512, 512
att = torch.randn(n, K, device=dev, generator=g)
wg = torch.randn(4 * H, K, device=dev, generator=g) * 0.05
gates_h = torch.randn(n, 4 * H, device=dev, generator=g)
gates_c = torch.randn(n, 4 * H, device=dev, generator=g)
emb = torch.randn(8519, 4 * H, device=dev, generator=g)
it = torch.randint(0, 8519, (n,), device=dev, generator=g)
c = torch.randn(n, H, device=dev, generator=g)
perm = MSDA.gate_permutation(H, dev)
ap, wpp = MSDA.split_rows(att), MSDA.split_rows(wg[perm].contiguous())
ghp, gcp, embp = gates_h[:, perm].contiguous(), gates_c[:, perm].contiguous(), emb[:, perm].contiguous()
gates = att.double() @ wg.double().t() + gates_h.double() + gates_c.double() + emb.double()[it]
i_, f_, g_, o_ = gates.chunk(4, 1)
c_ref = torch.sigmoid(f_) * c.double() + torch.sigmoid(i_) * torch.tanh(g_)
h_ref = torch.sigmoid(o_) * torch.tanh(c_ref)
for nprod in (3, 1):
    L.gvl_f16_products(nprod)
    h1, c1 = MSDA.gemm_f16x3_lstm(ap, wpp, ghp, gcp, embp, it, c)
    t = timed(lambda: MSDA.gemm_f16x3_lstm(ap, wpp, ghp, gcp, embp, it, c))
    print(f"gate product + cell x{nprod}: {t:6.1f} us, |h - ref| max {(h1.double() - h_ref).abs().max().item():.2e}, "
          f"|c - ref| max {(c1.double() - c_ref).abs().max().item():.2e}")
L.gvl_f16_products(3)

# k_lin: 4800 x 512 -> 512 and -> 2048 (relu)
for (R, K, N) in ((4800, 512, 512), (4800, 512, 2048), (4800, 2048, 512), (3008, 512, 768)):
    x = torch.randn(R, K, device=dev, generator=g)
    lin = torch.nn.Linear(K, N).to(dev)
    W = layers.Weights([(lin.weight, lin.bias)])
    am = layers.row_absmax(x)[0]
    ref = x.double() @ lin.weight.double().t() + lin.bias.double()
    for nprod in (3, 1):
        L.gvl_f16_products(nprod)
        out = torch.empty(R, N, device=dev)

        def run():
            layers.linear(x, W, [layers.seg(0, out, am)])
        run()
        err = (out.double() - ref).abs().max().item()
        print(f"k_lin {R}x{K}x{N} x{nprod}: {timed(run):6.1f} us, max err {err:.2e}")
    L.gvl_f16_products(3)
