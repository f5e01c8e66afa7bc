#!/usr/bin/env python3
"""Dev tool: wall time (with device sync) of the stages of the train step at bench shapes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from gvl_amd.config import make_opt
from gvl_amd.pdvc import build
from gvl_amd.parallel import TrainStep

dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
tr = TrainStep(model, criterion, opt)
dt = synth_batch(16, 100, 512, opt.vocab_size, 3, dev)
for _ in range(3):
    tr(dt)
acc = {}
def mark(name, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0) * 1e3
    return t1
N = 10
for _ in range(N):
    torch.cuda.synchronize(); t = time.perf_counter()
    tr.buckets.zero(); t = mark("zero_grad", t)
    out, loss = model(dt, criterion, None, "queries"); t = mark("forward(all)", t)
    wd = criterion.weight_dict
    final = sum(loss[k] * wd[k] for k in loss.keys() if k in wd); t = mark("loss sum", t)
    final.backward(); t = mark("backward", t)
    tr.buckets.finish()
    torch.nn.utils.clip_grad_norm_(tr.params, opt.grad_clip); t = mark("clip", t)
    tr.optimizer.step(); t = mark("adam", t)
print({k: round(v / N, 2) for k, v in acc.items()}, "sum", round(sum(acc.values()) / N, 2))
