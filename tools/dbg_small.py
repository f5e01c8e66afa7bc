import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA
dev = torch.device("cuda:0")
for R in (160, 40, 4, 129):
    x = torch.randn(R, 512, device=dev)
    xp = MSDA.split_rows(x); torch.cuda.synchronize(); print("split ok", R, flush=True)
    for N in (2560, 2048, 8518):
        w = torch.randn(N, 512, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        wp = MSDA.split_rows(w); torch.cuda.synchronize()
        o = MSDA.gemm_f16x3(xp, wp, b); torch.cuda.synchronize(); print(" gemm ok", R, N, float((o - torch.nn.functional.linear(x, w, b)).abs().max()), flush=True)
        p = MSDA.gemm_f16x3_argmax(xp, wp, b); torch.cuda.synchronize(); print(" argmax gemm ok", flush=True)
        tok, lp = MSDA.row_argmax_lse_partials(p); torch.cuda.synchronize(); print(" partials ok", bool((tok == o.argmax(1)).all()), flush=True)
