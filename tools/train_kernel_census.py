#!/usr/bin/env python3
"""Dev tool: how many kernels does each stage / autograd node of ONE eager train step launch (bench shapes)?
In the hipGraph-replayed step every kernel node costs a few microseconds regardless of its size, so the census of
launches -- not their durations -- says where the 20 ms go."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile, record_function

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch                      # noqa: E402
from gvl_amd.config import make_opt                # noqa: E402
from gvl_amd.parallel import TrainStep             # noqa: E402
from gvl_amd.pdvc import build                     # noqa: E402

dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
tr = TrainStep(model, criterion, opt, capturable=True)
dt = synth_batch(16, 100, 512, opt.vocab_size, 3, dev)
for _ in range(3):
    tr(dt)


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def inner(*a, **k):
        with record_function("stage:" + label):
            return fn(*a, **k)
    setattr(obj, name, inner)


wrap(model.base_encoder, "forward", "base_encoder")
wrap(model.transformer, "forward_encoder", "encoder")
wrap(model.transformer, "forward_decoder", "decoder")
wrap(model, "caption_prediction", "captioner_fwd")
wrap(criterion, "forward", "criterion+matcher")

torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    with record_function("stage:zero_grad"):
        tr.buckets.zero()
    with record_function("stage:forward_other"):
        out, loss = model(dt, criterion, None, "queries")
        wd = criterion.weight_dict
        final = sum(loss[k] * wd[k] for k in loss.keys() if k in wd)
    with record_function("stage:backward"):
        final.backward()
    with record_function("stage:clip"):
        tr.buckets.finish()
        torch.nn.utils.clip_grad_norm_(tr.params, opt.grad_clip)
    with record_function("stage:adam"):
        tr.optimizer.step()
    torch.cuda.synchronize()

evs = list(prof.profiler.kineto_results.events())
launches = [e for e in evs if "aunch" in e.name() and "Kernel" in e.name()]
stages = [e for e in evs if e.name().startswith("stage:")]
nodes = [e for e in evs if e.name().startswith("autograd::engine::evaluate_function: ")]


def inside(e, r):
    return r.start_ns() <= e.start_ns() <= r.start_ns() + r.duration_ns()


cnt = collections.Counter()
bwd = collections.Counter()
for l in launches:
    best = None
    for r in stages:
        if inside(l, r) and (best is None or r.duration_ns() < best.duration_ns()):
            best = r
    cnt[best.name() if best else "?"] += 1
    if best is not None and best.name() == "stage:backward":
        node = None
        for r in nodes:
            if inside(l, r) and (node is None or r.duration_ns() < node.duration_ns()):
                node = r
        bwd[node.name().split(": ")[1] if node else "?"] += 1
ops = [e for e in evs if e.name().startswith("aten::")]
detail = {st: collections.Counter() for st in sys.argv[1:]}
for l in launches:
    for st in detail:
        rng = [r for r in stages if r.name() == "stage:" + st and inside(l, r)]
        if rng:
            op = None
            for o in ops:                              # outermost aten op containing the launch
                if inside(l, o) and inside(o, rng[0]) and (op is None or o.duration_ns() > op.duration_ns()):
                    op = o
            detail[st][op.name() if op else "?"] += 1
for st, c in detail.items():
    print(f"aten ops launching kernels inside stage {st}:")
    for k, v in c.most_common(40):
        print(f"  {k:40s} {v:5d}")
print("kernel launches in one eager train step:", len(launches))
for k, v in cnt.most_common():
    print(f"  {k:28s} {v:6d}")
print("backward, by autograd node (top 30):")
for k, v in bwd.most_common(30):
    print(f"  {k:44s} {v:6d}")
