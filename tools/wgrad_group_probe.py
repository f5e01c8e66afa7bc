#!/usr/bin/env python3
"""Dev tool: the weight gradients of one encoder / decoder layer's Linears, one gvl_wgrad_f16x3_f32 launch (+ reduce) each against
one gvl_wgrad_group_f16x3_f32 launch (+ one reduce) for all of them -- event-timed over 50 repetitions, and the largest
difference of the results."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA      # noqa: E402
from gvl_amd import layers as L                                # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, R, shapes in (("encoder layer", 3008, [(512, 512), (256, 512), (512, 512), (512, 512), (512, 512)]),
                        ("decoder layer", 4800, [(512, 512), (256, 512), (512, 512), (512, 512), (512, 512), (1024, 512), (512, 512),
                                                 (512, 512)])):
    items = []
    for N, K in shapes:
        dy, x = torch.randn(R, N, device=dev), torch.randn(R, K, device=dev)
        items.append((dy, x, L.row_absmax(dy)[0], L.row_absmax(x)[0], torch.empty(N, K, device=dev), torch.empty(N, device=dev)))
    single = [(torch.empty_like(it[4]), torch.empty_like(it[5])) for it in items]

    def run_single():
        for it, (gw, gb) in zip(items, single):
            MSDA.wgrad(it[0], it[1], it[2], it[3], grad_w=gw, grad_b=gb)

    def run_group():
        MSDA.wgrad_group(items)
    for fn in (run_single, run_group):
        for _ in range(5):
            fn()
    torch.cuda.synchronize()
    res = {}
    for tag, fn in (("single", run_single), ("group", run_group), ("single2", run_single), ("group2", run_group)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[tag] = e0.elapsed_time(e1) / 50 * 1e3
    err = max(float((it[4] - gw).abs().max() / gw.abs().max()) for it, (gw, gb) in zip(items, single))
    errb = max(float((it[5] - gb).abs().max() / gb.abs().max()) for it, (gw, gb) in zip(items, single))
    print(f"{name} (R = {R}, {len(shapes)} Linears): one launch pair each {res['single']:.1f} / {res['single2']:.1f} us | grouped "
          f"{res['group']:.1f} / {res['group2']:.1f} us | max relative difference dW {err:.2e} db {errb:.2e}")
