#!/usr/bin/env python3
"""Dev tool: where the device-to-device memcpys (hipMemcpyAsync: aten::copy_ of contiguous same-dtype tensors) of one eager
training step come from: torch profiler with stacks, grouped by the op / autograd node that issued the copy."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rotating_batches                 # noqa: E402
from gvl_amd.config import make_opt                # noqa: E402
from gvl_amd.parallel import TrainStep, _PaddedBatch   # noqa: E402
from gvl_amd.pdvc import build                     # noqa: E402
from gvl_amd.tuning import enable_tuned_gemms      # noqa: E402

enable_tuned_gemms()
dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
batches = rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
step = TrainStep(model, criterion, opt, world_size=1)
pb = _PaddedBatch(batches[1], 16, 24, 96)
pb.load(batches[1])
for _ in range(3):
    step(pb.dt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step(pb.dt)
    torch.cuda.synchronize()
ev = prof.events()
copies = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA and "Memcpy" in e.name]
print("device memcpys in one step:", len(copies), "total us", sum(e.device_time for e in copies))
# parent CPU op of each memcpy: the innermost CPU event whose time range covers the memcpy's launch (correlation by time)
cpu = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
count = collections.Counter()
for e in cpu:
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy") and e.cpu_parent is not None:
        p = e.cpu_parent
        chain = [e.name]
        while p is not None and len(chain) < 4:
            chain.append(p.name)
            p = p.cpu_parent
        count[" <- ".join(chain)] += 1
for k, n in count.most_common(40):
    print(f"{n:5d}  {k}")
