"""Dev tool: gvl_col_sum_f32 against torch.sum(0) on the bias-gradient shapes of the train step (graph-replayed)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402


def timed(fn, iters=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for R, C in ((4800, 512), (4800, 2048), (4800, 256), (3008, 512), (3008, 2048), (1600, 512), (4800, 128), (96, 2576), (1056, 8519), (1056, 2576)):
    x = torch.randn(R, C, device="cuda")
    want = x.double().sum(0)
    got = MSDA.col_sum(x)
    err = (got.double() - want).abs().max().item()
    t_new = timed(lambda: MSDA.col_sum(x))
    t_lib = timed(lambda: x.sum(0))
    print(f"R={R:5d} C={C:5d}  col_sum {t_new:6.2f} us   torch.sum(0) {t_lib:6.2f} us   max err {err:.2e}", flush=True)
