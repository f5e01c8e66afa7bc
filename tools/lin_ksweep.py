#!/usr/bin/env python3
"""Dev tool: gvl_linear_f16x3_f32 at R = 4800, N in {64, 512, 2048} over K: fixed cost per launch vs cost per K stage."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import layers as L   # noqa: E402
from lin_bench import timeit      # noqa: E402

dev = "cuda:0"
for N in (64, 512, 2048):
    for K in (64, 128, 256, 512, 1024, 2048):
        R = 4800
        x = torch.randn(R, K, device=dev)
        w, b = torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
        W = L.Weights([(w, b)])
        am, _ = L.row_absmax(x)
        out = torch.empty(R, N, device=dev)
        segs = [L.seg(0, out, am)]
        t = timeit(lambda: L.linear(x, W, segs))
        print(f"N={N:5d} K={K:5d} stages={K // 32:3d}  {t:8.2f} us   {t / (K // 32):6.3f} us/stage")
