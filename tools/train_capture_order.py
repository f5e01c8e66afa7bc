"""Dev tool: is a GraphedTrainStep captured LATER in a process slower than one captured first?  (bench.py saw 10.4 ms for
the fixed 3-event layout on a second trainer against 8.8 ms in a process of its own)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rotating_batches, synth_batch
from gvl_amd.config import make_opt
from gvl_amd.parallel import GraphedTrainStep
from gvl_amd.pdvc import build
from gvl_amd.tuning import enable_tuned_gemms

enable_tuned_gemms()
dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
rot = rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
fixed = [synth_batch(16, 100, opt.feature_dim, opt.vocab_size, 3, dev, seed=1)]


def run(tag, batches, **kw):
    t = GraphedTrainStep(model, criterion, opt, world_size=1, **kw)
    for dt in batches:
        t(dt)
    for i in range(5):
        t(batches[i % len(batches)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30):
        t(batches[i % len(batches)])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    print(f"{tag}: {ms:6.2f} ms | captures {t.captures} | capacity {t.capacity.slots, t.capacity.cap_len, t.capacity.pair_rows}")
    return t


order = sys.argv[1] if len(sys.argv) > 1 else "fixed_first"
if order == "fixed_first":
    a = run("fixed (first trainer) ", fixed)
    b = run("rotating (second)     ", rot)
    c = run("fixed again (third)   ", fixed)
elif order == "rot_first":
    a = run("rotating (first)      ", rot)
    del a
    torch.cuda.empty_cache()
    b = run("fixed (second, first deleted)", fixed)
else:
    a = run("rotating, preset capacity (first)", rot, max_gt=10, max_cap_len=24, max_events=96)
