#!/usr/bin/env python3
"""Dev tool: the temporal backward at T = 512 (S = 960: level 0 in global memory, chunked queries), fp32 vs bf16 storage,
encoder (Lq = 960) and decoder (Lq = 300) shapes: kernel times from the library's dispatch stamps + the last chunk's phases."""
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
lens = [512, 256, 128, 64]
S = sum(lens)
tsh, lsi = make_level_tensors(lens, dev)
sh2 = temporal_shapes_2d(tsh, lsi)
B = int(os.environ.get("B", 16))
for dt in (torch.float32, torch.bfloat16):
    value = torch.randn(B, S, 8, 64, device=dev).to(dt)
    for name, Q, rd in (("enc Lq=960", S, 1), ("dec Lq=300", 300, 2)):
        proj = torch.randn(B, Q, 256, device=dev).to(dt)
        ref = torch.rand(B, Q, 4, rd, device=dev) * (0.5 if rd == 2 else 1.0)
        gout = torch.randn(B, Q, 512, device=dev).to(dt)
        for _ in range(3):
            MSDA.msda1d_fused_backward(value, sh2, lsi, proj, ref, gout, 4, 4, need_ref_grad=True)
        torch.cuda.synchronize()
        lib.gvl_msda_debug_stamps(buf.data_ptr())
        MSDA.profile_enable(True)
        for _ in range(5):
            MSDA.msda1d_fused_backward(value, sh2, lsi, proj, ref, gout, 4, 4, need_ref_grad=True)
        torch.cuda.synchronize()
        MSDA.profile_enable(False)
        lib.gvl_msda_debug_stamps(None)
        per = {}
        for tag, ma, mb, us in MSDA.profile_collect():
            per.setdefault(tag, []).append(us)
        nwg = min(256, 2 * B * 8)
        s_ = buf.view(-1, 4)[4096:4096 + nwg].cpu().numpy().astype(np.int64)
        us = (s_ - s_[:, 0:1]) / 100.0
        print(f"{str(dt):15s} {name}: " + " + ".join(f"{k} {np.median(v):.1f}" for k, v in per.items()) + " us | last chunk: "
              f"staged {np.mean(us[:, 1]):.2f}, phase 1 {np.mean(us[:, 2] - us[:, 1]):.2f}, phase 2 {np.mean(us[:, 3] - us[:, 2]):.2f}")
