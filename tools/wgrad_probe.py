#!/usr/bin/env python3
"""Dev tool: gvl_wgrad_f16x3_f32 against fp64 and against the library's `dy.t().mm(x)` + `dy.sum(0)` at the train step's shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA   # noqa: E402
from gvl_amd import layers as L                             # noqa: E402
from tools.lin_bench import timeit                          # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
for R, N, K in [(4800, 512, 512), (3008, 512, 512), (4800, 1536, 512), (4800, 256, 512), (3008, 1536, 512), (2208, 8520, 512),
                (4800, 512, 2048), (300, 512, 512), (4800, 64, 512)]:
    dy = torch.randn(R, N, device=dev) * torch.rand(R, 1, device=dev) * 1e-3
    x = torch.randn(R, K, device=dev)
    am_dy, _ = L.row_absmax(dy)
    am_x, _ = L.row_absmax(x)
    gw, gb = MSDA.wgrad(dy, x, am_dy, am_x)
    ref = dy.double().t().mm(x.double())
    refb = dy.double().sum(0)
    lib = dy.t().mm(x)
    den = ref.abs().max().item()
    e_own, e_lib = (gw.double() - ref).abs().max().item() / den, (lib.double() - ref).abs().max().item() / den
    r_own, r_lib = ((gw.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(), ((lib.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    eb = (gb.double() - refb).abs().max().item() / refb.abs().max().item()
    gw2, gb2 = gw.clone(), gb.clone()
    MSDA.wgrad(dy, x, am_dy, am_x, grad_w=gw2, grad_b=gb2, accumulate=True)
    acc_ok = torch.allclose(gw2, 2 * gw, rtol=1e-5, atol=1e-6 * gw.abs().max().item()) and torch.allclose(gb2, 2 * gb, rtol=1e-5, atol=1e-6 * gb.abs().max().item())
    t_own = timeit(lambda: MSDA.wgrad(dy, x, am_dy, am_x, grad_w=gw, grad_b=gb))
    t_lib = timeit(lambda: (dy.t().mm(x), dy.sum(0)))
    print(f"R={R:5d} N={N:5d} K={K:5d}  own {t_own:7.2f} us  lib(mm+sum) {t_lib:7.2f} us   max err/max: own {e_own:.2e} lib {e_lib:.2e}  "
          f"rms rel: own {r_own:.2e} lib {r_lib:.2e}  bias err {eb:.1e}  accumulate {'ok' if acc_ok else 'WRONG'}")
