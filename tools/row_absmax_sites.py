#!/usr/bin/env python3
"""Dev census: who still asks for a row-maxima launch (gvl_row_absmax_f32) in one eager training step at cfg A -- the producers on
the path leave the maxima behind themselves; every remaining call is an operand some other producer wrote."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch                      # noqa: E402
from gvl_amd import layers as L                    # noqa: E402
from gvl_amd.config import make_opt                # noqa: E402
from gvl_amd.parallel import TrainStep             # noqa: E402
from gvl_amd.pdvc import build                     # noqa: E402

dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
criterion = criterion.to(dev)
step = TrainStep(model, criterion, opt, world_size=1)
dt = synth_batch(16, 100, opt.feature_dim, opt.vocab_size, 3, dev)
for _ in range(2):
    step(dt)
sites = collections.Counter()
orig = L.row_absmax


def counted(x, pos=None, want_x=True):
    fr = [f for f in traceback.extract_stack()[:-1] if "gvl_amd" in f.filename]
    tag = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-6:])
    sites[(tag, tuple(x.shape))] += 1
    return orig(x, pos, want_x)


L.row_absmax = counted
step(dt)
torch.cuda.synchronize()
print(f"row_absmax launches in one training step: {sum(sites.values())}")
for (tag, shape), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"  {n:3d}  {str(shape):18s} {tag}")
