"""Dev probe: the hand-written fp32 MFMA projection kernel (gvl_proj_f32) against the library GEMM behind F.linear --
parity and GPU time inside a hipGraph of 50 calls (no host launch gaps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from gvl_amd import MultiScaleDeformableAttention as MSDA
from gvl_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device("cuda:0")


def in_graph_us(fn, n=50):
    for _ in range(3):
        fn()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * n)


for R in (4800, 3008, 37):
    x = torch.randn(R, 512, device=dev)
    w = torch.randn(256, 512, device=dev) * 0.05
    b = torch.randn(256, device=dev)
    ref64 = (x.double() @ w.double().t() + b.double())
    mine, lib = MSDA.proj_linear(x, w, b), F.linear(x, w, b)
    e_mine = float((mine.double() - ref64).abs().max()); e_lib = float((lib.double() - ref64).abs().max())
    t_mine = in_graph_us(lambda: MSDA.proj_linear(x, w, b)); t_lib = in_graph_us(lambda: F.linear(x, w, b))
    print(f"R={R}: hand-written {t_mine:6.2f} us ({2*R*512*256/t_mine/1e6:5.1f} TFLOP/s, max err vs fp64 {e_mine:.2e}) | "
          f"library {t_lib:6.2f} us ({2*R*512*256/t_lib/1e6:5.1f} TFLOP/s, err {e_lib:.2e})")
