#!/usr/bin/env python3
"""Dev tool: per-workgroup stamps of the sampling launch of EACH decoder layer inside an eager eval forward (the first layer's launch
follows the value projection of all layers + one small launch since the first layer's self-attention block is kept across forwards;
the second layer's follows its own projection product)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch                      # noqa: E402
from gvl_amd import _lib, layers as L              # noqa: E402
from gvl_amd.config import make_opt                # noqa: E402
from gvl_amd.pdvc import build                     # noqa: E402

dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, device="cuda", eval_disable_captioning=True)
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).eval()
dt = synth_batch(16, 100, 512, opt.vocab_size, 3, dev)
buf = torch.zeros(2 * 4 * 4096, dtype=torch.int64, device=dev)
lib = _lib.lib()
orig = L._msda
calls = []


def stamped(att, value, proj, ref, shapes2d, lsi, B, Lq, arena):
    calls.append(Lq)
    want = os.environ["WHICH"]
    idx = sum(1 for q in calls if q == 300)
    if Lq == 300 and str(idx) == want:
        lib.gvl_msda_debug_stamps(buf.data_ptr())
        out = orig(att, value, proj, ref, shapes2d, lsi, B, Lq, arena)
        lib.gvl_msda_debug_stamps(None)
        return out
    return orig(att, value, proj, ref, shapes2d, lsi, B, Lq, arena)


L._msda = stamped
with torch.no_grad():
    for _ in range(3):
        calls.clear()
        model(dt, None, None, "queries", eval_mode=True)
    for which in ("1", "2", "1", "2"):
        os.environ["WHICH"] = which
        for rep in range(3):
            calls.clear()
            buf.zero_()
            model(dt, None, None, "queries", eval_mode=True)
            torch.cuda.synchronize()
            s = buf.view(-1, 4)[:256].cpu().numpy().astype(np.int64)
            us = (s - s[:, 0].min()) / 100.0
            print(f"decoder layer {int(which) - 1}: WG start skew max {us[:, 0].max():5.2f} | staging mean {np.mean(us[:, 1] - us[:, 0]):5.2f} max "
                  f"{np.max(us[:, 1] - us[:, 0]):5.2f} | loop mean {np.mean(us[:, 2] - us[:, 1]):5.2f} | last WG done at {us[:, 2].max():5.2f} us")
