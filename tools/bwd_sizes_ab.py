#!/usr/bin/env python3
"""Dev tool: the fused temporal backward at other launch sizes than cfg A (bench.kernel_probe shapes: B = 8 / 32 / 64 / 256 at
T = 100, which the query-chunked k_bwd_t1d_d64 serves, and T = 200) -- median dispatch time and the kernel that served it.
Run once with the shipped library and once with GVL_LIB_PATH pointing at a timing build to compare."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import _lib                                                    # noqa: E402
from gvl_amd import MultiScaleDeformableAttention as MSDA                  # noqa: E402
from gvl_amd.deformable_transformer import make_level_tensors              # noqa: E402
from gvl_amd.ops.modules.ms_deform_attn import temporal_shapes_2d          # noqa: E402

dev = torch.device("cuda:0")
for B, T, Q in ((8, 100, 300), (32, 100, 300), (64, 100, 300), (256, 100, 300), (16, 200, 375), (32, 200, 300)):
    lens = [T]
    for _ in range(3):
        lens.append((lens[-1] - 1) // 2 + 1)
    S = sum(lens)
    tsh, lsi = make_level_tensors(lens, dev)
    sh2 = temporal_shapes_2d(tsh, lsi)
    g = torch.Generator(device=dev).manual_seed(7)
    value = torch.randn(B, S, 8, 64, device=dev, generator=g)
    proj = torch.randn(B, Q, 256, device=dev, generator=g)
    ref = torch.rand(B, Q, 4, 1, device=dev, generator=g)
    gout = torch.randn(B, Q, 512, device=dev, generator=g)
    for _ in range(3):
        MSDA.msda1d_fused_backward(value, sh2, lsi, proj, ref, gout, 4, 4)
    torch.cuda.synchronize()
    MSDA.profile_enable(True)
    MSDA.profile_collect()
    for _ in range(20):
        MSDA.msda1d_fused_backward(value, sh2, lsi, proj, ref, gout, 4, 4)
    torch.cuda.synchronize()
    MSDA.profile_enable(False)
    per = {}
    for tag, ma, mb, us in MSDA.profile_collect():
        per.setdefault(tag, []).append(us)
    nbytes = 4 * B * (2 * S * 512 + 6 * Q * 8 * 16 + Q * 512)
    tot = sum(float(np.median(v)) for v in per.values())
    print(f"B={B:3d} T={T} Q={Q}: {_lib.lib().gvl_msda_last_kernel().decode()}  " + " + ".join(f"{k} {np.median(v):.1f}" for k, v in per.items())
          + f" us  -> {nbytes / tot / 1e6 / 8:.3f} of 8 TB/s")
