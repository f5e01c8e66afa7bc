"""Probe: fp32 product on fp16 MFMA over a 2-way split with the residual scaled by 2^11 (3 partial products), rows scaled
to [1, 2) by a power of two.  Error vs fp64 and time against the library fp32 GEMM.  Also: does the fp16 MFMA path keep
subnormal inputs?"""
import torch

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)


def split2(x):
    """x (R, K) fp32 -> hi, lo' fp16, row exponent scale (R,) fp32 with x = scale * (hi + 2^-11 lo')"""
    m = x.abs().amax(1, keepdim=True).clamp_min(1e-30)
    e = torch.floor(torch.log2(m))
    s = torch.exp2(-e)
    xs = x * s
    hi = xs.to(torch.float16)
    hi = torch.where(hi.abs() < 2.0 ** -14, torch.zeros_like(hi), hi)        # no subnormal hi parts
    lo = ((xs - hi.float()) * 2048.0).to(torch.float16)
    lo = torch.where(lo.abs() < 2.0 ** -14, torch.zeros_like(lo), lo)
    return hi, lo, (1.0 / s).squeeze(1)


def timeit(fn, n=20):
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


# subnormal check: one fp16 subnormal times 1.0
a = torch.zeros(32, 32, device=dev, dtype=torch.float16)
b = torch.zeros(32, 32, device=dev, dtype=torch.float16)
a[0, 0] = 3e-6
b[0, 0] = 1.0
print("fp16 subnormal through mm (expect ~3e-6 if kept):", float(torch.mm(a, b.t(), out_dtype=torch.float32)[0, 0]))

for R, K, N, kind in ((4800, 512, 8518, "randn"), (4800, 512, 8518, "wide"), (4800, 512, 2560, "randn"), (4800, 1024, 2048, "randn")):
    x = torch.randn(R, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.05
    if kind == "wide":                                                 # elements spread over 12 orders of magnitude
        x = x * torch.exp2(torch.randint(-20, 20, (R, K), device=dev, generator=g).float())
        w = w * torch.exp2(torch.randint(-20, 20, (N, K), device=dev, generator=g).float())
    ref = x.double() @ w.double().t()
    lib = x @ w.t()
    xh, xl, xs = split2(x)
    wh, wl, ws = split2(w)
    main = torch.mm(xh, wh.t(), out_dtype=torch.float32)
    cross = torch.mm(torch.cat([xh, xl], 1), torch.cat([wl, wh], 1).t(), out_dtype=torch.float32)
    oa = (main + cross * (1.0 / 2048.0)) * xs[:, None] * ws[None, :]
    xa = torch.cat([xh * 2048.0, xh, xl], 1).contiguous()
    wa = torch.cat([wh, wl, wh], 1).contiguous()
    ob = torch.mm(xa, wa.t(), out_dtype=torch.float32) * (1.0 / 2048.0) * xs[:, None] * ws[None, :]
    o1 = main * xs[:, None] * ws[None, :]
    scale = float((x.abs().double() @ w.abs().double().t()).max())
    print(f"R={R} K={K} N={N} {kind}: sum|a||b| max {scale:.3g}")
    for name, o in (("fp32 lib", lib), ("fp16x3 two acc", oa), ("fp16x3 one acc", ob), ("fp16x1", o1)):
        d = (o.double() - ref)
        print(f"   {name}: max err vs fp64 {float(d.abs().max()):.3e}  rms {float(d.pow(2).mean().sqrt()):.3e}")
    fl = 2.0 * R * K * N
    t_lib = timeit(lambda: x @ w.t())
    t3 = timeit(lambda: torch.mm(xa, wa.t(), out_dtype=torch.float32))
    t1 = timeit(lambda: torch.mm(xh, wh.t(), out_dtype=torch.float32))
    print(f"   time: fp32 lib {t_lib:.1f} us ({fl / t_lib / 1e6:.0f} TF) | fp16x3 (K'=3K) {t3:.1f} us ({3 * fl / t3 / 1e6:.0f} TF fp16) | fp16x1 {t1:.1f} us")
