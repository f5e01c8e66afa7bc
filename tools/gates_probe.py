"""gvl_gemm_f16x3_gates_f32 at cfg A's token-step shape (n = 4800, H = 512, K = 512 + 512): launch time from HIP events; with a
GVL_G_STAMPS build the two workgroups of tile 0 print their phases.   python tools/gates_probe.py [--reps 50]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n, H, Ka, V = 4800, 512, 512, 8519
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)           # noqa: E731
    att, h_prev = rnd(n, Ka), torch.tanh(rnd(n, H))
    w_cat = MSDA.split_rows((rnd(4 * H, H + Ka) * 0.04).contiguous())
    gates_c, emb, c = rnd(n, 4 * H), rnd(V, 4 * H), rnd(n, H)
    it = torch.randint(0, V, (n,), device=dev, generator=g)
    ap_, hp = MSDA.split_rows(att), MSDA.split_rows(h_prev)
    for _ in range(5):
        MSDA.gemm_f16x3_gates(ap_, hp, w_cat, gates_c, emb, it, c)
    torch.cuda.synchronize()
    for rnd_ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            MSDA.gemm_f16x3_gates(ap_, hp, w_cat, gates_c, emb, it, c)
        e1.record()
        torch.cuda.synchronize()
        print(f"round {rnd_}: {e0.elapsed_time(e1) * 1e3 / a.reps:.1f} us per launch", flush=True)


if __name__ == "__main__":
    main()
