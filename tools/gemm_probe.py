#!/usr/bin/env python3
"""Dev probe: the captioner's three per-token fp32 GEMMs under hipBLASLt default vs PyTorch TunableOp."""
import os, sys, time, torch
dev = torch.device("cuda:0")
shapes = {"logits  (4800x512)x(512x8518)": (4800, 512, 8518), "h_cat   (4800x512)x(512x2560)": (4800, 512, 2560),
          "att     (4800x512)x(512x2048)": (4800, 512, 2048), "ffn     (4800x512)x(512x512)": (4800, 512, 512)}
def bench(tag):
    for name, (m, k, n) in shapes.items():
        x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
        for _ in range(5): torch.nn.functional.linear(x, w, b)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): torch.nn.functional.linear(x, w, b)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print(f"{tag:9s} {name}: {us:8.1f} us  {2*m*k*n/us/1e6:6.1f} TFLOP/s")
bench("default")
torch.cuda.tunable.enable(True); torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(200); torch.cuda.tunable.set_max_tuning_iterations(30)
t0 = time.time(); bench("tunable"); print("tuning took", time.time() - t0, "s")
bench("tuned")
print(torch.cuda.tunable.get_results()[:8])
