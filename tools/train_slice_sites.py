#!/usr/bin/env python3
"""Dev tool: forward source lines of the Slice / Select / MaskedFill / View-clone backward nodes of one training step
(each costs a zero fill + a copy + usually an accumulation in the backward): autograd anomaly mode keeps the forward stack."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rotating_batches                 # noqa: E402
from gvl_amd.config import make_opt                # noqa: E402
from gvl_amd.parallel import TrainStep, _PaddedBatch   # noqa: E402
from gvl_amd.pdvc import build                     # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
batches = rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
step = TrainStep(model, criterion, opt, world_size=1)
pb = _PaddedBatch(batches[1], 16, 24, 96)
pb.load(batches[1])
step(pb.dt)
with torch.autograd.set_detect_anomaly(True, check_nan=False):
    final, loss = step._forward_loss(pb.dt)
seen, stack, count = set(), [final.grad_fn], collections.Counter()
while stack:
    n = stack.pop()
    if n is None or n in seen:
        continue
    seen.add(n)
    name = type(n).__name__
    if name in ("SliceBackward0", "SelectBackward0", "MaskedFillBackward0", "IndexBackward0", "ExpandBackward0",
                "RepeatBackward0", "CatBackward0", "StackBackward0"):
        tb = n.metadata.get("traceback_", [])
        site = "?"
        for line in reversed(tb):
            if ROOT in line and "/tools/" not in line:
                site = line.strip().split("\n")[0].replace(ROOT + "/", "")
                break
        count[(name, site)] += 1
    for nxt, _ in n.next_functions:
        stack.append(nxt)
for (name, site), c in sorted(count.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{c:4d}  {name:22s} {site}")
