#!/usr/bin/env python3
"""Dev tool: every non-view aten op of ONE eager training step (bench shapes): the forward's by gvl_amd source line
(TorchDispatchMode + Python stack), the backward's / optimizer's by op name and by the autograd node that issued it."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rotating_batches                 # noqa: E402
from gvl_amd.config import make_opt                # noqa: E402
from gvl_amd.parallel import TrainStep             # noqa: E402
from gvl_amd.pdvc import build                     # noqa: E402
from gvl_amd.parallel import _PaddedBatch          # noqa: E402
from gvl_amd.tuning import enable_tuned_gemms      # noqa: E402

VIEWS = {"view", "reshape", "_unsafe_view", "transpose", "permute", "slice", "select", "expand", "unsqueeze", "squeeze",
         "t", "detach", "alias", "as_strided", "unbind", "split", "chunk", "split_with_sizes", "_reshape_alias", "unfold",
         "lift_fresh", "is_same_size", "sym_size", "sym_stride", "sym_numel", "view_as_real", "movedim", "narrow", "flatten"}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASE = ["forward"]


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()
        self.order = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in VIEWS:
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if fr.filename.startswith(ROOT) and "/tools/" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
                    break
            shape = ""
            for a in args:
                if isinstance(a, torch.Tensor):
                    shape = "x".join(map(str, a.shape))
                    break
            self.count[(PHASE[0], site, name, shape)] += 1
            if PHASE[0] == "backward" and not name.startswith(("empty", "new_empty")):
                shapes = ["x".join(map(str, a.shape)) for a in args if isinstance(a, torch.Tensor)]
                rest = [a for a in args if (isinstance(a, int) and not isinstance(a, bool)) or (isinstance(a, (list, tuple)) and all(isinstance(e, int) for e in a))][:2]
                if name in ('stack', 'cat') and args and isinstance(args[0], (list, tuple)):
                    shapes = ['x'.join(map(str, t_.shape)) for t_ in args[0]]
                self.order.append(f"{name:24s} {' | '.join(shapes):60s} {rest} {site}")
        return func(*args, **(kwargs or {}))


enable_tuned_gemms()
dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
batches = rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
step = TrainStep(model, criterion, opt, world_size=1)
dts = []
assert model.supports_padded_targets(criterion, False, batch=16, slots=16)
for b in batches[:2]:
    pb = _PaddedBatch(b, 16, 24, 96)
    pb.load(b)
    dts.append(pb.dt)
for dt in dts:
    step(dt)
torch.cuda.synchronize()
dt = dts[1]
with Sites() as s:
    step.buckets.zero()
    final, loss = step._forward_loss(dt)
    PHASE[0] = "backward"
    final.backward()
    PHASE[0] = "optimizer"
    step.buckets.finish()
    torch.nn.utils.clip_grad_norm_([p for p in step.params if p.grad is not None], opt.grad_clip)
    step.optimizer.step()
tot = collections.Counter()
for (ph, site, name, shape), n in s.count.items():
    tot[ph] += n
print("non-view aten ops of one training step:", dict(tot))
fwd = collections.Counter()
for (ph, site, name, shape), n in s.count.items():
    if ph == "forward":
        fwd[site.split(":")[0]] += n
for f, n in fwd.most_common():
    print(f"  forward {f:50s} {n:5d}")
print("forward by site:")
key = lambda kv: (kv[0][1].split(':')[0], int(kv[0][1].split(':')[1]) if ':' in kv[0][1] else 0)   # noqa: E731
for (ph, site, name, shape), n in sorted(((k, v) for k, v in s.count.items() if k[0] == "forward"), key=key):
    print(f"  {site:55s} {name:28s} {shape:22s} {n:4d}")
for phase in ("backward", "optimizer"):
    print(phase, "by op and site:")
    agg = collections.Counter()
    for (ph, site, name, shape), n in s.count.items():
        if ph == phase:
            agg[(site, name)] += n
    for (site, name), n in agg.most_common(70):
        print(f"  {site:55s} {name:28s} {n:4d}")
    print(phase, "by op, shape (top 60):")
    agg = collections.Counter()
    for (ph, site, name, shape), n in s.count.items():
        if ph == phase:
            agg[(name, shape)] += n
    for (name, shape), n in agg.most_common(60):
        print(f"  {name:28s} {shape:24s} {n:4d}")
print("backward in order (no empties):")
for i, l in enumerate(s.order):
    print(f"  {i:4d} {l}")
