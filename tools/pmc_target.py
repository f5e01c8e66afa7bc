#!/usr/bin/env python3
"""Target of the rocprofv3 --pmc passes (tools/pmc_run.sh): the fused temporal forward / backward at ONE configuration,
launch order fixed -- per shape 1 warm-up + 3 measured launches, shapes in the order fwd enc, fwd dec, bwd enc, bwd dec.
usage: pmc_target.py <T> <f32|bf16> [amax]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA          # noqa: E402
from gvl_amd.deformable_transformer import make_level_tensors      # noqa: E402
from gvl_amd.ops.modules.ms_deform_attn import temporal_shapes_2d  # noqa: E402

T, dt = int(sys.argv[1]), (torch.bfloat16 if sys.argv[2] == "bf16" else torch.float32)
amax = len(sys.argv) > 3 and sys.argv[3] == "amax"
dev = torch.device("cuda:0")
lens = [T]
for _ in range(3):
    lens.append((lens[-1] - 1) // 2 + 1)
S, B, Qd = sum(lens), 16, 300
tsh, lsi = make_level_tensors(lens, dev)
sh2 = temporal_shapes_2d(tsh, lsi)
g = torch.Generator(device=dev).manual_seed(0)
value = torch.randn(B, S, 8, 64, device=dev, generator=g).to(dt)
for kind in ("fwd", "bwd"):
    for Q, rd in ((S, 1), (Qd, 2)):
        proj = torch.randn(B, Q, 256, device=dev, generator=g).to(dt)
        ref = torch.rand(B, Q, 4, rd, device=dev, generator=g) * (0.5 if rd == 2 else 1.0)
        gout = torch.randn(B, Q, 512, device=dev, generator=g).to(dt)
        am = torch.zeros(B * Q, device=dev) if (amax and kind == "fwd") else None
        for _ in range(4):
            if kind == "fwd":
                MSDA.msda1d_fused_forward(value, sh2, lsi, proj, ref, 4, 4, amax_out=am)
            else:
                MSDA.msda1d_fused_backward(value, sh2, lsi, proj, ref, gout, 4, 4, need_ref_grad=True)
        torch.cuda.synchronize()
