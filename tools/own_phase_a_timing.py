#!/usr/bin/env python3
"""Dev tool (round 4, needs a library built with GVL_BUILD_DEFS=-DGVL_PHASE_TIMING): cycles wave 0 of every workgroup spends in
the blocks of a phase-A pass of k_bwd_t1d_own (coefficients | 16 sample steps | two reduce-scatters | epilogue + stores)."""
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
lib = _lib.lib()
buf = torch.zeros(2 * 4 * 4096, dtype=torch.int64, device=dev)
B = 16
for T, Q, rd in ((200, 375, 1), (512, 960, 1), (512, 300, 2)):
    lens = [T]
    for _ in range(3):
        lens.append((lens[-1] - 1) // 2 + 1)
    S = sum(lens)
    tsh, lsi = make_level_tensors(lens, dev)
    sh2 = temporal_shapes_2d(tsh, lsi)
    g = torch.Generator(device=dev).manual_seed(3)
    value = torch.randn(B, S, 8, 64, device=dev, generator=g)
    proj = torch.randn(B, Q, 256, device=dev, generator=g)
    ref = torch.rand(B, Q, 4, rd, device=dev, generator=g) * (0.5 if rd == 2 else 1.0)
    gout = torch.randn(B, Q, 512, device=dev, generator=g)
    for _ in range(3):
        MSDA.msda1d_fused_backward(value, sh2, lsi, proj, ref, gout, 4, 4, need_ref_grad=True)
    torch.cuda.synchronize()
    buf.zero_()
    lib.gvl_msda_debug_stamps(buf.data_ptr())
    MSDA.msda1d_fused_backward(value, sh2, lsi, proj, ref, gout, 4, 4, need_ref_grad=True)
    torch.cuda.synchronize()
    lib.gvl_msda_debug_stamps(None)
    x = buf.view(-1, 4)[4096 + 2048:4096 + 2048 + 256].cpu().numpy().astype(np.float64)
    npass = -(-((Q + 1) // 2) // 64)                 # passes of wave 0
    print(f"T={T} Lq={Q} ({lib.gvl_msda_last_kernel().decode()}): wave 0 ran {npass} passes; cycles per pass: "
          f"coefficients {x[:, 0].mean() / npass:6.0f} | sample steps {x[:, 1].mean() / npass:6.0f} | reduce-scatters {x[:, 2].mean() / npass:6.0f} | "
          f"epilogue {x[:, 3].mean() / npass:6.0f} | sum {x.sum(1).mean() / npass:6.0f}")
