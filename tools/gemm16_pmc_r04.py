"""three launches of the vocabulary product (fused argmax form, m16 kernel) with 3 and with 1 fp16 products per fp32 product
(for rocprofv3 --pmc passes; tools/gemm16_pmc_r04.sh)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import _lib                                                    # noqa: E402
from gvl_amd import MultiScaleDeformableAttention as MSDA                  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(4800, 512, device=dev, generator=g)
w = torch.randn(8518, 512, device=dev, generator=g) * 0.05
b = torch.randn(8518, device=dev, generator=g)
xp, wp = MSDA.split_rows(x), MSDA.split_rows(w)
for n in (3, 1):
    _lib.lib().gvl_f16_products(n)
    for _ in range(3):
        MSDA.gemm_f16x3_argmax(xp, wp, b)
torch.cuda.synchronize()
