#!/usr/bin/env python3
"""Dev tool: every non-view, non-allocation aten op of ONE eager eval forward (bench shapes) by gvl_amd source line
(TorchDispatchMode + Python stack): what torch still launches around the hand-written kernels."""
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
from gvl_amd.tuning import enable_tuned_gemms      # noqa: E402

SKIP = {"view", "reshape", "_unsafe_view", "transpose", "permute", "slice", "select", "expand", "unsqueeze", "squeeze", "t",
        "detach", "alias", "as_strided", "unbind", "split", "chunk", "split_with_sizes", "_reshape_alias", "unfold", "lift_fresh",
        "is_same_size", "sym_size", "sym_stride", "sym_numel", "movedim", "narrow", "flatten", "empty", "empty_like",
        "new_empty", "empty_strided", "new_empty_strided", "_local_scalar_dense"}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in SKIP:
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
            self.count[(site, name, shape)] += 1
        return func(*args, **(kwargs or {}))


enable_tuned_gemms()
dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).eval()
batches = rotating_batches(2, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
with torch.no_grad():
    for b in batches:
        model(b, criterion, None, opt.transformer_input_type, eval_mode=True)
    torch.cuda.synchronize()
    with Sites() as s:
        model(batches[1], criterion, None, opt.transformer_input_type, eval_mode=True)
print("aten ops of one eval forward (allocations and views not counted):", sum(s.count.values()))
by_op = collections.Counter()
for (site, name, shape), n in s.count.items():
    by_op[name] += n
print(" ".join(f"{k}:{v}" for k, v in by_op.most_common()))
key = lambda kv: (kv[0][0].split(':')[0], int(kv[0][0].split(':')[1]) if ':' in kv[0][0] else 0)   # noqa: E731
for (site, name, shape), n in sorted(s.count.items(), key=key):
    print(f"  {site:55s} {name:24s} {shape:22s} {n:4d}")
