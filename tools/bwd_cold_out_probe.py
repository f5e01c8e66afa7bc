#!/usr/bin/env python3
"""Dev tool (round 4): is the bf16 backward slow when its OUTPUT buffers are cold?  The fused backward writes grad_proj as
32-byte pieces per (query, head) in bf16 (64-byte pieces in fp32).  Back to back the allocator hands out the same block every
call (lines resident in L2 / Infinity Cache); here the results of the last N calls are kept alive so that every launch
writes memory it has never touched (N x ~15 MB >> 256 MB Infinity Cache when N = 40)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA                  # noqa: E402
from gvl_amd.deformable_transformer import make_level_tensors              # noqa: E402
from gvl_amd.ops.modules.ms_deform_attn import temporal_shapes_2d          # noqa: E402

dev = torch.device("cuda:0")
B = 16
for T, Q, rd in ((100, 300, 2), (100, 188, 1), (512, 300, 2)):
    lens = [T]
    for _ in range(3):
        lens.append((lens[-1] - 1) // 2 + 1)
    S = sum(lens)
    tsh, lsi = make_level_tensors(lens, dev)
    sh2 = temporal_shapes_2d(tsh, lsi)
    for dt in (torch.float32, torch.bfloat16):
        g = torch.Generator(device=dev).manual_seed(3)
        value = torch.randn(B, S, 8, 64, device=dev, generator=g).to(dt)
        proj = torch.randn(B, Q, 256, device=dev, generator=g).to(dt)
        ref = torch.rand(B, Q, 4, rd, device=dev, generator=g) * (0.5 if rd == 2 else 1.0)
        gout = torch.randn(B, Q, 512, device=dev, generator=g).to(dt)
        for keep in (0, 60):
            held = []
            for _ in range(3):
                MSDA.msda1d_fused_backward(value, sh2, lsi, proj, ref, gout, 4, 4, need_ref_grad=False)
            torch.cuda.synchronize()
            MSDA.profile_enable(True)
            for _ in range(40):
                r = MSDA.msda1d_fused_backward(value, sh2, lsi, proj, ref, gout, 4, 4, need_ref_grad=False)
                if keep:
                    held.append(r)
            torch.cuda.synchronize()
            MSDA.profile_enable(False)
            us = [e[3] for e in MSDA.profile_collect()]
            print(f"T={T} Lq={Q} {str(dt)[6:]:8s} outputs {'cold (60 calls kept)' if keep else 'reused':22s} median {np.median(us):6.2f} us  "
                  f"p90 {np.percentile(us, 90):6.2f}  max {max(us):6.2f}", flush=True)
            del held
            torch.cuda.empty_cache()
