#!/usr/bin/env python3
"""Dev tool (round 4): the temporal backward at the long-video shapes (T = 512: encoder Lq = 960, decoder Lq = 300) and at
cfg A, fp32 and bf16 storage, row-ownership form (k_bwd_t1d_own) against the query-chunked form it replaces
(GVL_MSDA_BWD_OWN=0: k_bwd_t1d_d64<loop> + k_sum_partials).  Kernel times = the library's dispatch stamps (median of 10
back-to-back launches); phases = per-workgroup wall-clock stamps {start, slab staged, phase A done, phase B done}."""
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
B = int(os.environ.get("B", 16))
HBM = 8.0e12


def bwd_bytes(S, Q, vb):
    return B * (2 * vb * S * 512 + 4 * 6 * Q * 128 + vb * Q * 512)


def run(T, Q, rd, dt, tag, env):
    lens = [T]
    for _ in range(3):
        lens.append((lens[-1] - 1) // 2 + 1)
    S = sum(lens)
    tsh, lsi = make_level_tensors(lens, dev)
    sh2 = temporal_shapes_2d(tsh, lsi)
    g = torch.Generator(device=dev).manual_seed(3)
    value = torch.randn(B, S, 8, 64, device=dev, generator=g).to(dt)
    proj = torch.randn(B, Q, 256, device=dev, generator=g).to(dt)
    ref = torch.rand(B, Q, 4, rd, device=dev, generator=g) * (0.5 if rd == 2 else 1.0)
    gout = torch.randn(B, Q, 512, device=dev, generator=g).to(dt)
    for k, v in env.items():
        os.environ[k] = v
    lib.gvl_reload_env()                                     # (the library caches its switches)
    try:
        for _ in range(3):
            MSDA.msda1d_fused_backward(value, sh2, lsi, proj, ref, gout, 4, 4, need_ref_grad=True)
        torch.cuda.synchronize()
        kern = lib.gvl_msda_last_kernel().decode()
        buf.zero_()
        lib.gvl_msda_debug_stamps(buf.data_ptr())
        MSDA.profile_enable(True)
        for _ in range(10):
            MSDA.msda1d_fused_backward(value, sh2, lsi, proj, ref, gout, 4, 4, need_ref_grad=True)
        torch.cuda.synchronize()
        MSDA.profile_enable(False)
        lib.gvl_msda_debug_stamps(None)
    finally:
        for k in env:
            del os.environ[k]
        lib.gvl_reload_env()
    per = {}
    for tg, ma, mb, us in MSDA.profile_collect():
        per.setdefault(tg, []).append(us)
    tot = sum(float(np.median(v)) for v in per.values())
    nwg = min(256, 2 * B * 8)
    s_ = buf.view(-1, 4)[4096:4096 + nwg].cpu().numpy().astype(np.int64)
    us = (s_ - s_[:, 0:1]) / 100.0
    vb = 2 if dt == torch.bfloat16 else 4
    nb = bwd_bytes(S, Q, vb)
    ph = ""
    if kern == "k_bwd_t1d_own":
        ph = (f" | staged {np.mean(us[:, 1]):.1f}, phase A {np.mean(us[:, 2] - us[:, 1]):.1f} (max {np.max(us[:, 2] - us[:, 1]):.1f}), "
              f"phase B {np.mean(us[:, 3] - us[:, 2]):.1f} (max {np.max(us[:, 3] - us[:, 2]):.1f}); g0/g1 end "
              f"{np.mean(us[:nwg // 2, 3]):.1f}/{np.mean(us[nwg // 2:, 3]):.1f}")
        x = buf.view(-1, 4)[4096 + 1024:4096 + 1024 + nwg].cpu().numpy().astype(np.int64)
        ph += (f" | last chunk: coefficient pass {np.mean(x[:, 1] - x[:, 0]) / 100:.1f}, scan + sort {np.mean(x[:, 2] - x[:, 1]) / 100:.1f}, "
               f"gather {np.mean(s_[:, 3] - x[:, 2]) / 100:.1f} (max {np.max(s_[:, 3] - x[:, 2]) / 100:.1f})")
    print(f"T={T:4d} Lq={Q:4d} {str(dt)[6:]:9s} {tag:10s} {kern:22s} " + " + ".join(f"{k} {np.median(v):.1f}" for k, v in per.items())
          + f" us = {tot:6.1f} us -> {nb / tot / 1e6 / HBM * 1e12:.3f} of 8 TB/s" + ph, flush=True)


for dt in (torch.float32, torch.bfloat16):
    for T, Q, rd in ((512, 960, 1), (512, 300, 2), (512, 100, 2), (200, 375, 1), (100, 300, 2), (100, 188, 1)):
        run(T, Q, rd, dt, "default", {})
        if T > 100:
            run(T, Q, rd, dt, "OWN=0", {"GVL_MSDA_BWD_OWN": "0"})
if os.environ.get("SWEEP"):
    for qc in (128, 192, 256, 320, 384):
        run(512, 960, 1, torch.float32, f"qc={qc}", {"GVL_MSDA_BWD_OWN_QC": str(qc)})
if os.environ.get("BIGB"):
    B = int(os.environ["BIGB"])
    for T, Q, rd in ((512, 960, 1), (512, 300, 2), (100, 300, 2), (100, 188, 1), (200, 300, 2)):
        run(T, Q, rd, torch.float32, "default", {})
        run(T, Q, rd, torch.float32, "OWN=0", {"GVL_MSDA_BWD_OWN": "0"})
