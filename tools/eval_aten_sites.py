#!/usr/bin/env python3
"""Dev tool: every non-view aten op of ONE eager eval forward (bench shapes) with the gvl_amd source line that issued it
(TorchDispatchMode + Python stack): where the small PyTorch launches outside the hand-written kernels come from."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rotating_batches                 # noqa: E402
from gvl_amd.config import make_opt                # noqa: E402
from gvl_amd.pdvc import build                     # noqa: E402
from gvl_amd.targets import PaddedTargets          # noqa: E402

VIEWS = {"view", "reshape", "_unsafe_view", "transpose", "permute", "slice", "select", "expand", "unsqueeze", "squeeze",
         "t", "detach", "alias", "as_strided", "unbind", "split", "chunk", "split_with_sizes", "_reshape_alias", "unfold",
         "lift_fresh", "is_same_size", "sym_size", "sym_stride", "sym_numel", "view_as_real", "movedim", "narrow", "flatten"}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in VIEWS:
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if fr.filename.startswith(ROOT) and "/tools/" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
                    break
            self.count[(site, name)] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda",
               eval_disable_captioning="--no-captioner" in sys.argv)
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).eval()
batches = rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
dt = dict(batches[0])
pt = PaddedTargets(16, 16, 0, dev)
pt.load(dt)
dt["_gvl_targets"] = pt
with torch.no_grad():
    model(dt, criterion, None, "queries", eval_mode=True)
    torch.cuda.synchronize()
    with Sites() as s:
        model(dt, criterion, None, "queries", eval_mode=True)
by_file = collections.Counter()
for (site, name), n in s.count.items():
    by_file[site.split(":")[0]] += n
print("non-view aten ops of one eval forward:", sum(s.count.values()))
for f, n in by_file.most_common():
    print(f"  {f:50s} {n:5d}")
print("by site:")
for (site, name), n in sorted(s.count.items(), key=lambda kv: (kv[0][0].split(':')[0], int(kv[0][0].split(':')[1]) if ':' in kv[0][0] else 0)):
    print(f"  {site:55s} {name:28s} {n:4d}")
