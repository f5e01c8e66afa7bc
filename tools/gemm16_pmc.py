"""one launch of each gvl_gemm_f16x3 form at the path's shapes (for rocprofv3 --pmc passes)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(4800, 512, device=dev, generator=g)
xp = MSDA.split_rows(x)
for N in (8518, 2560):
    w = torch.randn(N, 512, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    wp = MSDA.split_rows(w)
    for _ in range(3):
        MSDA.gemm_f16x3(xp, wp, b)
        if N == 8518:
            MSDA.gemm_f16x3_argmax(xp, wp, b)
torch.cuda.synchronize()
