#!/usr/bin/env python3
"""Dev tool: back-to-back timing of gvl_linear_f16x3_f32 / gvl_layer_norm_rows_f32 against the library calls they replace,
at the shapes of the cfg A eval forward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import layers as L   # noqa: E402

dev = "cuda:0"


def timeit(fn, n=50, reps=5):
    """device time per call: n calls captured in one hipGraph (no host launch cost between them), best of `reps` replays"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / n)
    return best


def main():
    torch.manual_seed(0)
    rows = []
    for name, R, K, N, kw in [("enc value|proj", 3008, 512, 768, dict(two=True)), ("enc out_proj+res", 3008, 512, 512, dict(res=True)),
                              ("enc ffn1 relu", 3008, 512, 2048, dict(relu=True, amax=True)), ("enc ffn2+res", 3008, 2048, 512, dict(res=True)),
                              ("dec in_proj", 4800, 512, 1536, dict(two=True)), ("dec proj", 4800, 512, 256, dict(addend=True)),
                              ("dec out_proj+res", 4800, 512, 512, dict(res=True)), ("dec ffn1 relu", 4800, 512, 2048, dict(relu=True, amax=True)),
                              ("dec ffn2+res", 4800, 2048, 512, dict(res=True)), ("dec values x3", 3008, 512, 1536, dict()),
                              ("mlp0", 4800, 512, 512, dict(relu=True, amax=True)), ("mlp2 (N=2)", 4800, 512, 64, dict())]:
        x, pos = torch.randn(R, K, device=dev), torch.randn(R, K, device=dev)
        w, b = torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
        W = L.Weights([(w, b)])
        am, amp = L.row_absmax(x, pos)
        out, res, am2 = torch.empty(R, N, device=dev), torch.randn(R, N, device=dev), torch.zeros(R, device=dev)
        if kw.get("two"):
            half = (N // 2) // 64 * 64
            segs = [L.seg(0, out[:, :half], am), L.seg(half, out[:, half:], amp, addend=True)]
        else:
            segs = [L.seg(0, out, amp if kw.get("addend") else am, resid=res if kw.get("res") else None,
                          relu=kw.get("relu", False), amax_out=am2 if kw.get("amax") else None, addend=kw.get("addend", False))]
        t_own = timeit(lambda: L.linear(x, W, segs, a2=pos))
        t_xcd = timeit(lambda: L.linear(x, W, segs, a2=pos, flags=L.LIN_XCD_COLUMNS)) if N == 512 else float("nan")
        t_lib = timeit(lambda: torch.nn.functional.linear(x, w, b))
        rows.append((name, R, K, N, t_own, t_xcd, t_lib, 2.0 * R * K * N / t_own * 1e-6))
    print(f"{'product':20s} {'R':>5s} {'K':>5s} {'N':>5s} {'own us':>8s} {'xcd us':>8s} {'lib us':>8s} {'TFLOP/s(fp32 eq)':>16s}")
    for r in rows:
        print(f"{r[0]:20s} {r[1]:5d} {r[2]:5d} {r[3]:5d} {r[4]:8.2f} {r[5]:8.2f} {r[6]:8.2f} {r[7]:16.1f}")
    for R in (3008, 4800):
        x, pos = torch.randn(R, 512, device=dev), torch.randn(R, 512, device=dev)
        norm = torch.nn.LayerNorm(512).to(dev)
        print(f"layer_norm R={R}: own {timeit(lambda: L.layer_norm(x, norm, pos=pos)):.2f} us (with both row maxima), "
              f"torch {timeit(lambda: norm(x)):.2f} us; row_absmax {timeit(lambda: L.row_absmax(x, pos)):.2f} us; "
              f"torch add {timeit(lambda: x + pos):.2f} us")


if __name__ == "__main__" and "--mha" not in sys.argv and "--skinny" not in sys.argv:
    main()


def mha():
    B, Q, H = 16, 300, 8
    qkv = torch.randn(B * Q, 3 * H * 64, device=dev)
    am = torch.zeros(B * Q, device=dev)
    t = qkv.view(B, Q, 3, H, 64).permute(2, 0, 3, 1, 4)
    keep = torch.ones(B, Q, dtype=torch.bool, device=dev)
    print(f"mha_core B=16 Q=300 H=8: own {timeit(lambda: L.mha_core(qkv, B, Q, H, keep, am)):.2f} us, "
          f"torch SDPA (+ mask) {timeit(lambda: torch.nn.functional.scaled_dot_product_attention(t[0], t[1], t[2], attn_mask=keep[:, None, None, :])):.2f} us")


if __name__ == "__main__" and "--mha" in sys.argv:
    mha()


def skinny():
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    for M, K, N in ((192, 512, 2576), (192, 512, 2048), (192, 2048, 512), (192, 2576, 512), (96, 512, 2576)):
        x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05
        p = MSDA.skinny_pack(w)
        out = torch.empty(M, N, device=dev)
        wt = w.t()
        # a dependent chain (each product waits for the previous one), as in the token loop: latency, not throughput
        t_own = timeit(lambda: MSDA.skinny_gemm(x, p, out))
        t_lib = timeit(lambda: torch.mm(x, wt, out=out))
        print(f"skinny M={M} K={K} N={N}: own {t_own:.2f} us, library {t_lib:.2f} us")


if __name__ == "__main__" and "--skinny" in sys.argv:
    skinny()
