"""Time gvl_gemm_f16x3_f32 (+ the split of the activation operand) against the fp32 library GEMM and against the
library's fp16 GEMM on K-concatenated planes, on the three products of one captioner token step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for R, K, N in ((4800, 512, 8518), (4800, 512, 2560), (4800, 512, 2048), (2208, 512, 8518)):
    x = torch.randn(R, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    ref = x.double() @ w.double().t() + b.double()
    xp, wp = MSDA.split_rows(x), MSDA.split_rows(w)
    out = torch.empty(R, N, device=dev)
    mine = MSDA.gemm_f16x3(xp, wp, b, out=out)
    lib = torch.nn.functional.linear(x, w, b)
    (xh, xl), (wh, wl) = xp.dense(), wp.dense()
    xa = torch.cat([xh * 2048.0, xh, xl], 1).contiguous()
    wa = torch.cat([wh, wl, wh], 1).contiguous()
    t_mine = timeit(lambda: MSDA.gemm_f16x3(xp, wp, b, out=out))
    t_split = timeit(lambda: MSDA.split_rows(x, out=xp))
    t_lib = timeit(lambda: torch.nn.functional.linear(x, w, b))
    t_lib16 = timeit(lambda: torch.mm(xa, wa.t(), out_dtype=torch.float32))
    fl = 2.0 * R * K * N
    print(f"R={R} K={K} N={N}: gvl_gemm_f16x3 {t_mine:7.1f} us ({3 * fl / t_mine / 1e6:5.0f} TF fp16, {fl / t_mine / 1e6:4.0f} TF fp32-equivalent)"
          f" + split {t_split:.1f} us | library fp32 {t_lib:7.1f} us | library fp16 K'=3K {t_lib16:7.1f} us | "
          f"rms err vs fp64: mine {float((mine.double() - ref).pow(2).mean().sqrt()):.2e} lib {float((lib.double() - ref).pow(2).mean().sqrt()):.2e}")
