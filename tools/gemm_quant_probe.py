#!/usr/bin/env python3
"""Dev tool: how does the time of the per-token captioner GEMMs depend on N (tile quantisation)?  TunableOp tunes
every shape, so each point is the best library kernel for that shape."""
import sys
import torch
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(100)
torch.cuda.tunable.set_max_tuning_iterations(20)
torch.cuda.tunable.set_filename("/tmp/gq.csv", False)
dev = torch.device("cuda:0")
M, K = 4800, 512


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


x = torch.randn(M, K, device=dev)
for N in [int(a) for a in sys.argv[1:]] or [8518, 8480, 8448, 8320, 8192, 7680, 38, 70, 2560, 2496, 2048, 512]:
    w = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    us = timeit(lambda: torch.nn.functional.linear(x, w, b))
    print(f"N={N:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
