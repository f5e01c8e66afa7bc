#!/usr/bin/env python3
"""Dev tool: where do the microseconds of k_fwd_t1d_d64 go IN SITU (inside an eager eval forward) and back-to-back?
Per-workgroup wall-clock stamps {start, slab staged, loop done} (gvl_msda_debug_stamps)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch                      # noqa: E402
from gvl_amd import _lib                           # noqa: E402
from gvl_amd.config import make_opt                # noqa: E402
from gvl_amd.pdvc import build                     # noqa: E402

dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, device="cuda", eval_disable_captioning=True)
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).eval()
dt = synth_batch(16, 100, 512, opt.vocab_size, 3, dev)
buf = torch.zeros(2 * 4 * 4096, dtype=torch.int64, device=dev)
lib = _lib.lib()


def report(tag):
    s = buf.view(-1, 4)[:256].cpu().numpy().astype(np.int64)
    t0 = s[:, 0].min()
    us = (s - t0) / 100.0                              # 100 MHz -> microseconds
    print(f"{tag}: WG start skew max {us[:, 0].max():5.2f} | staging (start->staged) mean {np.mean(us[:, 1] - us[:, 0]):5.2f} "
          f"max {np.max(us[:, 1] - us[:, 0]):5.2f} | loop mean {np.mean(us[:, 2] - us[:, 1]):5.2f} max {np.max(us[:, 2] - us[:, 1]):5.2f} "
          f"| last WG done at {us[:, 2].max():5.2f} us after the first WG started")


with torch.no_grad():
    for _ in range(3):
        model(dt, None, None, "queries", eval_mode=True)
    torch.cuda.synchronize()
    # in situ: stamp the LAST fused launch of a forward (= decoder layer 1 cross-attention, Lq = 300)
    lib.gvl_msda_debug_stamps(buf.data_ptr())
    model(dt, None, None, "queries", eval_mode=True)
    torch.cuda.synchronize()
    report("in situ, decoder launch ")
    # back-to-back on the same operands
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    from gvl_amd.deformable_transformer import make_level_tensors
    from gvl_amd.ops.modules.ms_deform_attn import temporal_shapes_2d
    tsh, lsi = make_level_tensors([100, 50, 25, 13], dev)
    sh2 = temporal_shapes_2d(tsh, lsi)
    value = torch.randn(16, 188, 8, 64, device=dev)
    proj = torch.randn(16, 300, 256, device=dev)
    ref = torch.rand(16, 300, 4, 2, device=dev) * 0.5
    for _ in range(5):
        MSDA.msda1d_fused_forward(value, sh2, lsi, proj, ref, 4, 4)
    torch.cuda.synchronize()
    report("back-to-back, same shape")
    lib.gvl_msda_debug_stamps(None)

    # ---- backward, back-to-back (decoder shape) ---------------------------------------------------------------
    gout = torch.randn(16, 300, 512, device=dev)
    lib.gvl_msda_debug_stamps(buf.data_ptr())
    for _ in range(3):
        MSDA.msda1d_fused_backward(value, sh2, lsi, proj, ref, gout, 4, 4, need_ref_grad=True)
    torch.cuda.synchronize()
    s_ = buf.view(-1, 4)[4096:4096 + 256].cpu().numpy().astype(np.int64)
    us = (s_ - s_[:, 0].min()) / 100.0
    print(f"backward, back-to-back: start skew max {us[:, 0].max():5.2f} | staging mean {np.mean(us[:, 1] - us[:, 0]):5.2f} | "
          f"phase 1 mean {np.mean(us[:, 2] - us[:, 1]):5.2f} | phase 2 mean {np.mean(us[:, 3] - us[:, 2]):5.2f} | "
          f"phase 2 done at max {us[:, 3].max():5.2f} us (phase 3 = the rest of the kernel)")
    # ---- backward, long video (cfg L encoder shape, B = 64): the stamps describe each workgroup's LAST query chunk ----
    from gvl_amd import _lib as _l
    tsh, lsi = make_level_tensors([512, 256, 128, 64], dev)
    sh2 = temporal_shapes_2d(tsh, lsi)
    Bq, Sq = 64, 960
    value = torch.randn(Bq, Sq, 8, 64, device=dev)
    proj = torch.randn(Bq, Sq, 256, device=dev)
    ref = torch.rand(Bq, Sq, 4, 1, device=dev)
    gout = torch.randn(Bq, Sq, 512, device=dev)
    for _ in range(3):
        MSDA.msda1d_fused_backward(value, sh2, lsi, proj, ref, gout, 4, 4, need_ref_grad=True)
    torch.cuda.synchronize()
    s_ = buf.view(-1, 4)[4096:4096 + 512].cpu().numpy().astype(np.int64)
    us = s_ / 100.0
    print(f"backward cfg L enc B=64, last chunk of each workgroup: staging mean {np.mean(us[:, 1] - us[:, 0]):5.2f} | "
          f"phase 1 mean {np.mean(us[:, 2] - us[:, 1]):5.2f} | phase 2 mean {np.mean(us[:, 3] - us[:, 2]):5.2f} | "
          f"kernel span {(us[:, 3].max() - us[:, 0].min()):7.1f} us over {len(us)} workgroups")
    lib.gvl_msda_debug_stamps(None)
