#!/usr/bin/env python3
"""Dev tool (round 4): same-run A/B of the fused temporal kernels under different environment switches.  For every shape the
variants are run in alternation (R rounds x 20 back-to-back launches, library dispatch stamps), so box / clock drift cancels.
usage: msda_ab_probe.py fwd|bwd  VAR=val[,VAR=val...] [VAR=val ...]      (the empty variant "-" = defaults)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import _lib                                                    # noqa: E402
from gvl_amd import MultiScaleDeformableAttention as MSDA                  # noqa: E402
from gvl_amd.deformable_transformer import make_level_tensors              # noqa: E402
from gvl_amd.ops.modules.ms_deform_attn import temporal_shapes_2d          # noqa: E402

kind = sys.argv[1]
variants = [("-" if v == "-" else v, {} if v == "-" else dict(kv.split("=") for kv in v.split(","))) for v in sys.argv[2:]] or [("-", {})]
dev = torch.device("cuda:0")
lib = _lib.lib()
B = int(os.environ.get("B", 16))
R = int(os.environ.get("ROUNDS", 5))
shapes = ((100, 300, 2), (100, 188, 1), (512, 300, 2), (512, 960, 1))
for dt in (torch.float32, torch.bfloat16) if os.environ.get("BF16") else (torch.float32,):
    for T, Q, rd in shapes:
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
        am = torch.zeros(B * Q, device=dev)

        def call():
            if kind == "bwd":
                MSDA.msda1d_fused_backward(value, sh2, lsi, proj, ref, gout, 4, 4, need_ref_grad=True)
            elif kind == "fwd_amax" and dt == torch.float32:
                MSDA.msda1d_fused_forward(value, sh2, lsi, proj, ref, 4, 4, amax_out=am)
            else:
                MSDA.msda1d_fused_forward(value, sh2, lsi, proj, ref, 4, 4)
        res = {name: [] for name, _ in variants}
        for r in range(R):
            for name, env in variants:
                os.environ.update(env)
                _lib.reload_env()                            # (the library caches its switches)
                try:
                    for _ in range(3):
                        call()
                    torch.cuda.synchronize()
                    MSDA.profile_enable(True)
                    for _ in range(20):
                        call()
                    torch.cuda.synchronize()
                    MSDA.profile_enable(False)
                finally:
                    for k_ in env:
                        del os.environ[k_]
                    _lib.reload_env()
                per = {}
                for tg, ma, mb, us in MSDA.profile_collect():
                    per.setdefault(tg, []).append(us)
                res[name].append(sum(float(np.median(v)) for v in per.values()))
        vb = 2 if dt == torch.bfloat16 else 4
        nb = B * ((2 if kind == "bwd" else 1) * vb * S * 512 + 4 * (6 if kind == "bwd" else 3) * Q * 128 + vb * Q * 512)
        print(f"{kind} T={T:4d} Lq={Q:4d} {str(dt)[6:]:8s} " + " | ".join(
            f"{name}: {np.median(v):6.2f} us (min {min(v):6.2f}) = {nb / np.median(v) / 8e6:.3f}" for name, v in res.items()), flush=True)
