#!/usr/bin/env python3
"""Dev tool: per-workgroup phase stamps of the temporal backward at the cfg A decoder / encoder shapes, back-to-back.
stamps = {start, own pass begins (slab staged; split form: foreign pass done), phase 1 done, phase 2 done}; kernel time from
the library's dispatch stamps.  GVL_MSDA_BWD_SPLIT=0 selects the query-split form (+ k_sum_partials)."""
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
tsh, lsi = make_level_tensors([100, 50, 25, 13], dev)
sh2 = temporal_shapes_2d(tsh, lsi)
value = torch.randn(16, 188, 8, 64, device=dev)
for name, Q, rd in (("dec (Lq=300, ref-dim 2)", 300, 2), ("enc (Lq=188, ref-dim 1)", 188, 1)):
    proj = torch.randn(16, Q, 256, device=dev)
    ref = torch.rand(16, Q, 4, rd, device=dev) * (0.5 if rd == 2 else 1.0)
    if os.environ.get("PROBE_SAME_ROW"):     # every sample of a level at one location: the lane = sample reads meet no bank conflicts
        proj[..., :128] = 0
        ref = torch.full_like(ref, 0.37)
    gout = torch.randn(16, Q, 512, device=dev)
    for _ in range(5):
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
    s_ = buf.view(-1, 4)[4096:4096 + 256].cpu().numpy().astype(np.int64)
    us = (s_ - s_[:, 0].min()) / 100.0
    print(f"{name}: kernels " + " + ".join(f"{k} {np.median(v):.2f}" for k, v in per.items()) + " us | "
          f"start skew max {us[:, 0].max():.2f} | start->own pass mean {np.mean(us[:, 1] - us[:, 0]):.2f} | "
          f"own pass mean {np.mean(us[:, 2] - us[:, 1]):.2f} max {np.max(us[:, 2] - us[:, 1]):.2f} | "
          f"phase 2 mean {np.mean(us[:, 3] - us[:, 2]):.2f} | phase 2 done at mean {np.mean(us[:, 3]):.2f} max {us[:, 3].max():.2f}")
    e3 = (buf.view(-1, 4)[4096 + 1024:4096 + 1024 + 256, 0].cpu().numpy().astype(np.int64) - s_[:, 0].min()) / 100.0
    if e3.max() > 0:
        print(f"    gather (phase 3) done at mean {e3.mean():.2f} max {e3.max():.2f}; per group: {e3[:128].mean():.2f} / {e3[128:].mean():.2f}")
    for g in (0, 1):
        u = us[g * 128:(g + 1) * 128]
        print(f"    workgroups g={g}: own pass begins {np.mean(u[:, 1]):.2f}, phase 1 done {np.mean(u[:, 2]):.2f}, phase 2 done {np.mean(u[:, 3]):.2f}")
