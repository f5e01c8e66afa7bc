import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gvl_amd
import bench
dev = torch.device("cuda:0")
for B in (16, 64, 256):
    for T in (100, 200, 512):
        if B * T > 256 * 200 and T == 512:
            continue
        f = bench.kernel_probe(dev, B, T=T, Q=300)
        b = bench.kernel_probe(dev, B, T=T, Q=300, backward=True)
        print(f"B={B:3d} T={T:3d}: fwd {f['kernel_us']:7.2f} us = {f['frac']:.3f} | bwd {b['kernel_us']:7.2f} us = {b['frac']:.3f}", flush=True)
