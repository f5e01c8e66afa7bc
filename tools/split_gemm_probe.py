"""Probe: fp32 product emulated by bf16 MFMA over a 3-way split of both operands (6 partial products, K' = 6 K) against
the library fp32 GEMM: time and error vs fp64.  Usage: python tools/split_gemm_probe.py"""
import time
import torch

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)


def split3(x):
    x0 = x.to(torch.bfloat16)
    r = x - x0.float()
    x1 = r.to(torch.bfloat16)
    r = r - x1.float()
    return x0, x1, r.to(torch.bfloat16)


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


for R, K, N in ((4800, 512, 8518), (4800, 512, 2560), (4800, 1024, 2048)):
    x = torch.randn(R, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.05
    ref = x.double() @ w.double().t()
    lib = x @ w.t()
    x0, x1, x2 = split3(x)
    w0, w1, w2 = split3(w)
    print(f"R={R} K={K} N={N}: split residual x {float((x - x0.float() - x1.float() - x2.float()).abs().max()):.2e}")
    xa6 = torch.cat([x0, x0, x1, x0, x1, x2], 1).contiguous()
    wa6 = torch.cat([w0, w1, w0, w2, w1, w0], 1).contiguous()
    xa3 = torch.cat([x0, x0, x1], 1).contiguous()
    wa3 = torch.cat([w0, w1, w0], 1).contiguous()
    try:
        o6 = torch.mm(xa6, wa6.t(), out_dtype=torch.float32)
        o3 = torch.mm(xa3, wa3.t(), out_dtype=torch.float32)
    except Exception as e:                                             # noqa: BLE001
        print("mm out_dtype unsupported:", repr(e)[:300])
        break
    for name, o in (("fp32 lib", lib), ("bf16x6", o6), ("bf16x3", o3)):
        print(f"   {name}: max err vs fp64 {float((o.double() - ref).abs().max()):.3e}  rms {float((o.double() - ref).pow(2).mean().sqrt()):.3e}")
    t_lib = timeit(lambda: x @ w.t())
    t6 = timeit(lambda: torch.mm(xa6, wa6.t(), out_dtype=torch.float32))
    t3 = timeit(lambda: torch.mm(xa3, wa3.t(), out_dtype=torch.float32))
    t1 = timeit(lambda: torch.mm(x0, w0.t(), out_dtype=torch.float32))
    tb = timeit(lambda: torch.mm(xa6, wa6.t()))
    fl = 2.0 * R * K * N
    print(f"   time: fp32 lib {t_lib:.1f} us ({fl / t_lib / 1e6:.0f} TF) | bf16x6 {t6:.1f} us ({6 * fl / t6 / 1e6:.0f} TF bf16) | "
          f"bf16x3 {t3:.1f} us | bf16x1 {t1:.1f} us | bf16x6 bf16-out {tb:.1f} us")
    ts = timeit(lambda: split3(x))
    print(f"   split3(x) eager: {ts:.1f} us")
