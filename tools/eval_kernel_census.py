#!/usr/bin/env python3
"""Dev tool: kernel launches of ONE eager eval forward (bench shapes), by stage and by aten op inside the decoding loop."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile, record_function

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch                      # noqa: E402
from gvl_amd.config import make_opt                # noqa: E402
from gvl_amd.pdvc import build                     # noqa: E402

dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).eval()
dt = synth_batch(16, 100, 512, opt.vocab_size, 3, dev)


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def inner(*a, **k):
        with record_function("stage:" + label):
            return fn(*a, **k)
    setattr(obj, name, inner)


wrap(model.base_encoder, "forward", "base_encoder")
wrap(model.transformer, "forward_encoder", "encoder")
wrap(model.transformer, "forward_decoder", "decoder")
wrap(model.caption_head[-1], "_decode_device", "decode_loop")
wrap(criterion, "forward", "criterion+matcher")
with torch.no_grad():
    for _ in range(2):
        model(dt, criterion, None, "queries", eval_mode=True)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        with record_function("stage:other"):
            model(dt, criterion, None, "queries", eval_mode=True)
        torch.cuda.synchronize()
evs = list(prof.profiler.kineto_results.events())
launches = [e for e in evs if "aunch" in e.name() and "Kernel" in e.name()]
stages = [e for e in evs if e.name().startswith("stage:")]
ops = [e for e in evs if e.name().startswith("aten::")]


def inside(e, r):
    return r.start_ns() <= e.start_ns() <= r.start_ns() + r.duration_ns()


cnt, loop = collections.Counter(), collections.Counter()
for l in launches:
    best = None
    for r in stages:
        if inside(l, r) and (best is None or r.duration_ns() < best.duration_ns()):
            best = r
    cnt[best.name() if best else "?"] += 1
    if best is not None and best.name() == "stage:decode_loop":
        op = None
        for o in ops:
            if inside(l, o) and inside(o, best) and (op is None or o.duration_ns() > op.duration_ns()):
                op = o
        loop[op.name() if op else "(library kernel)"] += 1
print("kernel launches in one eager eval forward:", len(launches))
for k, v in cnt.most_common():
    print(f"  {k:28s} {v:6d}")
print("inside the decoding loop, by aten op:")
for k, v in loop.most_common(30):
    print(f"  {k:36s} {v:6d}")
