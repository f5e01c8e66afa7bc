#!/usr/bin/env python3
"""Target of the rocprofv3 --pmc passes of tools/pmc_cap.sh: the captioner's attention kernel of one token step at the
bench shape (16 videos x 300 queries, T = 100 pyramid), plain (k_cap_attend) and with the coarse levels in LDS
(k_cap_attend_lds): 1 warm-up + 3 launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA          # noqa: E402
from gvl_amd.deformable_transformer import make_level_tensors      # noqa: E402
from gvl_amd.ops.modules.ms_deform_attn import temporal_shapes_2d  # noqa: E402

dev = torch.device("cuda:0")
lens = [100, 50, 25, 13]
S, B, Q, C = sum(lens), 16, 300, 512
tsh, lsi = make_level_tensors(lens, dev)
sh2 = temporal_shapes_2d(tsh, lsi)
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)          # noqa: E731
slab = rnd(B, S, 2 * C)
ref = torch.rand(B, Q, 4, 2, device=dev, generator=g) * 0.5
off_hs, h, w_off, att_h, aw = rnd(B, Q, 16), rnd(B * Q, C) * 0.3, rnd(16, C) * 0.05, rnd(B * Q, C), rnd(C) * 0.1
starts = [0, 100, 150, 175]
for host in (None, starts):
    for _ in range(4):
        MSDA.cap_attend(slab, sh2, lsi, ref, off_hs, h, w_off, att_h, aw, 0.1, 4, 4, planes=True, host_starts=host)
    torch.cuda.synchronize()
