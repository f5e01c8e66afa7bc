#!/usr/bin/env python3
"""Op-level micro-benchmark of the MSDA kernels (dev tool; bench.py is the contract benchmark).
usage: python tools/opbench.py [--B 16] [--T 100] [--Q 300] [--iters 200]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA, _lib  # noqa: E402


def levels(T, n=4):
    out = [T]
    for _ in range(n - 1):
        out.append((out[-1] - 1) // 2 + 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--Q", type=int, default=300)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--cold", action="store_true", help="evict L2 / Infinity Cache (write 1 GiB) before every profiled launch")
    ap.add_argument("--fresh", action="store_true", help="rewrite the kernel's inputs with another kernel right before every profiled launch")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, D, L, P = 8, 64, 4, 4
    lens = levels(a.T)
    S = sum(lens)
    shapes = torch.tensor([(1, x) for x in lens], dtype=torch.long, device=dev)
    lsi = torch.tensor(np.concatenate([[0], np.cumsum(lens)[:-1]]), dtype=torch.long, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    value = torch.randn(a.B, S, M, D, device=dev, generator=g)
    evict = torch.zeros(1 << 28, device=dev) if a.cold else None
    for name, Q in (("enc", S), ("dec", a.Q)):
        loc = torch.rand(a.B, Q, M, L, P, 2, device=dev, generator=g) * 1.5 - 0.25
        loc[..., 1] = 0.5
        aw = torch.softmax(torch.randn(a.B, Q, M, L * P, device=dev, generator=g), -1).view(a.B, Q, M, L, P)
        gout = torch.randn(a.B, Q, M * D, device=dev, generator=g)
        fbytes = 4 * a.B * (S * 512 + 3 * Q * M * L * P + Q * 512)
        bbytes = 4 * a.B * (2 * S * 512 + 6 * Q * M * L * P + Q * 512)
        for impl, code in (("generic", 1), ("fast", 2)):
            _lib.lib().gvl_msda_set_impl(code)
            for what, fn, nbytes in (
                    ("fwd", lambda: MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64), fbytes),
                    ("bwd", lambda: MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, aw, gout, 64), bbytes)):
                try:
                    for _ in range(10):
                        fn()
                except RuntimeError as e:
                    print(f"{name} {impl} {what}: not eligible ({e})")
                    continue
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                wall = e0.elapsed_time(e1) * 1e3 / a.iters
                MSDA.profile_enable(True)
                for _ in range(20):
                    if a.cold:
                        MSDA.profile_enable(False)
                        evict.add_(1.0)
                        MSDA.profile_enable(True)
                    fn()
                torch.cuda.synchronize()
                MSDA.profile_enable(False)
                per = {}
                for tag, ma, mb, t_us in MSDA.profile_collect():
                    per.setdefault(tag, []).append(t_us)
                ktxt = " + ".join(f"{k} {sorted(v)[len(v) // 2]:.2f}" for k, v in per.items())
                us = sum(sorted(v)[len(v) // 2] for v in per.values())
                print(f"{name:3s} Q={Q:4d} {impl:7s} {what}: wall {wall:7.2f} us/call | kernels {us:7.2f} us ({ktxt}) | "
                      f"alg {nbytes / 1e6:6.2f} MB -> {nbytes / us / 1e6:6.3f} TB/s = {nbytes / us / 1e6 / 8 * 100:5.1f}% of 8 TB/s")
    _lib.lib().gvl_msda_set_impl(0)
    # fused module path: proj = [offsets | logits], ref points
    for name, Q in (("enc", S), ("dec", a.Q)):
        proj = torch.randn(a.B, Q, 2 * M * L * P, device=dev, generator=g)
        ref = torch.rand(a.B, Q, L, 1, device=dev, generator=g)
        gout = torch.randn(a.B, Q, M * D, device=dev, generator=g)
        shapes._gvl_host = (np.array([(1, x) for x in lens], np.int64), np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64))
        fbytes = 4 * a.B * (S * 512 + 3 * Q * M * L * P + Q * 512)
        bbytes = 4 * a.B * (2 * S * 512 + 6 * Q * M * L * P + Q * 512)
        for what, fn, nbytes in (
                ("fwd", lambda: MSDA.msda1d_fused_forward(value, shapes, lsi, proj, ref, L, P), fbytes),
                ("bwd", lambda: MSDA.msda1d_fused_backward(value, shapes, lsi, proj, ref, gout, L, P), bbytes)):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            MSDA.profile_enable(True)
            for _ in range(20):
                if a.cold:
                    MSDA.profile_enable(False)
                    evict.add_(1.0)
                    MSDA.profile_enable(True)
                if a.fresh:
                    value.mul_(1.0); proj.mul_(1.0); ref.mul_(1.0); gout.mul_(1.0)
                fn()
            torch.cuda.synchronize()
            MSDA.profile_enable(False)
            per = {}
            for tag, ma, mb, t_us in MSDA.profile_collect():
                per.setdefault(tag, []).append(t_us)
            us = sum(sorted(v)[len(v) // 2] for v in per.values())
            ktxt = " + ".join(f"{k} {sorted(v)[len(v) // 2]:.2f}" for k, v in per.items())
            print(f"{name:3s} Q={Q:4d} fused   {what}: kernels {us:7.2f} us ({ktxt}) | alg {nbytes / 1e6:6.2f} MB -> "
                  f"{nbytes / us / 1e6 / 8 * 100:5.1f}% of 8 TB/s")


if __name__ == "__main__":
    main()
