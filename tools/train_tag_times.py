#!/usr/bin/env python3
"""Dev tool: mean duration per library kernel tag over a few EAGER train steps at the bench shapes (in-library
hipExtLaunchKernel stamps).   python tools/train_tag_times.py"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch                      # noqa: E402
from gvl_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402
from gvl_amd.config import make_opt                # noqa: E402
from gvl_amd.parallel import TrainStep             # noqa: E402
from gvl_amd.pdvc import build                     # noqa: E402
from gvl_amd.tuning import enable_tuned_gemms      # noqa: E402

enable_tuned_gemms()
dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
tr = TrainStep(model, criterion, opt, capturable=True)
dt = synth_batch(16, 100, 512, opt.vocab_size, 3, dev)
for _ in range(3):
    tr(dt)
torch.cuda.synchronize()
MSDA.profile_enable(True)
for _ in range(3):
    tr(dt)
torch.cuda.synchronize()
MSDA.profile_enable(False)
per = collections.defaultdict(list)
for tag, ma, mb, us in MSDA.profile_collect():
    per[tag].append(us)
for tag, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f"{tag:16s} n/step {len(v) / 3:6.1f}  mean {sum(v) / len(v):7.2f} us  per step {sum(v) / 3:8.1f} us")
