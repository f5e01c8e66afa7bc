"""The vocabulary product + fused argmax (gvl_gemm_f16x3_argmax_f32) on the form GVL_VOCAB_FORM selects ("m16" | "v" | unset = the
dispatch rule): tokens / log-probabilities against fp64 at a few shapes, then the launch time at (4800, 8518, 512).
    python tools/vocab_probe.py [--reps 200]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402


def check(R, V, K, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(R, K, device=dev, generator=g)
    w = torch.randn(V, K, device=dev, generator=g) * 0.05
    b = torch.randn(V, device=dev, generator=g)
    xp, wp = MSDA.split_rows(x), MSDA.split_rows(w)
    tok, lp = MSDA.row_argmax_lse_partials(MSDA.gemm_f16x3_argmax(xp, wp, b))
    ref = torch.log_softmax(x.double() @ w.double().t() + b.double(), 1)
    lp_ref, tok_ref = ref.max(1)
    bad = int((tok != tok_ref).sum())
    # a differing token is a near-tie iff the reference's log-probability at OUR token is within rounding of its maximum
    gap = float((lp_ref - ref.gather(1, tok[:, None])[:, 0]).max())
    err = float((lp.double() - lp_ref).abs().max())
    print(f"R={R} V={V} K={K}: tokens differing {bad} (largest fp64 gap {gap:.2e}), log-prob error {err:.2e}", flush=True)
    return bad == 0 or gap < 1e-5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--time-only", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    print("GVL_VOCAB_FORM =", os.environ.get("GVL_VOCAB_FORM"))
    ok = True
    for R, V, K, seed in [] if a.time_only else [(4800, 8518, 512, 1), (1600, 8518, 512, 2), (1030, 300, 128, 3), (2000, 70, 64 * 3, 4),
                          (3333, 5000, 1024, 5), (4801, 8519, 512, 6), (3613, 1657, 256, 7)]:
        ok &= check(R, V, K, dev, seed)
    R, V, K = 4800, 8518, 512
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(R, K, device=dev, generator=g)
    w = torch.randn(V, K, device=dev, generator=g) * 0.05
    b = torch.randn(V, device=dev, generator=g)
    xp, wp = MSDA.split_rows(x), MSDA.split_rows(w)
    for _ in range(20):
        p = MSDA.gemm_f16x3_argmax(xp, wp, b)
    torch.cuda.synchronize()
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            p = MSDA.gemm_f16x3_argmax(xp, wp, b)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        print(f"round {rnd}: {us:.1f} us per launch = {2 * 3 * R * V * K / us / 1e9:.3f} PFLOP/s of fp16 MFMA "
              f"({2 * 3 * R * V * K / us / 1e9 / 2.5:.3f} of 2.5)", flush=True)
    del p
    print("OK" if ok else "MISMATCH")


if __name__ == "__main__":
    main()
