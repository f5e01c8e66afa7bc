"""Dev tool: the teacher-forced loop's skinny products (96-320 rows) on k_gemm_f16x3 (four-wavefront form) against the
library GEMM: kernel durations from the library's stamps / torch events over many launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA
from gvl_amd.tuning import enable_tuned_gemms

enable_tuned_gemms()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for R, K, N in ((96, 512, 2576), (96, 512, 2048), (320, 512, 2576), (96, 2576, 512), (96, 2048, 512)):
    Kp = (K + 31) // 32 * 32
    x = torch.zeros(R, Kp, device=dev); x[:, :K] = torch.randn(R, K, device=dev, generator=g)
    w = torch.zeros(N, Kp, device=dev); w[:, :K] = torch.randn(N, K, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    xp, wp = MSDA.split_rows(x), MSDA.split_rows(w)
    out = torch.empty(R, N, device=dev)
    for _ in range(5):
        MSDA.gemm_f16x3(xp, wp, b, out=out)
    torch.cuda.synchronize()
    MSDA.profile_enable(1)
    for _ in range(30):
        MSDA.gemm_f16x3(xp, wp, b, out=out)
        MSDA.split_rows(x, out=xp)
    torch.cuda.synchronize()
    MSDA.profile_enable(False)
    rows = MSDA.profile_collect()
    mine = sorted(u for t, a, b_, u in rows if t == "gemm_f16x3")
    spl = sorted(u for t, a, b_, u in rows if t == "split_rows")
    wt = w[:, :K].t().contiguous()
    xs = x[:, :K].contiguous()
    for _ in range(5):
        torch.addmm(b, xs, wt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(gr, stream=side):
        for _ in range(50):
            o2 = torch.addmm(b, xs, wt)
    gr.replay(); torch.cuda.synchronize()
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    print(f"R={R} K={K} N={N}: k_gemm_f16x3 median {mine[len(mine) // 2]:6.2f} us (+ split {spl[len(spl) // 2]:.2f}) | library addmm "
          f"{e0.elapsed_time(e1) * 1e3 / 50:6.2f} us per call (50 back to back in a graph)")
