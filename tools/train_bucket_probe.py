#!/usr/bin/env python3
"""Dev tool: per-replay time of the captured train step under the two caption-width policies (grow: one graph at the
widest width seen; bucket: one graph per width bucket of 4 tokens): is a bucket graph slow by itself, or only right after
another graph ran?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rotating_batches                 # noqa: E402
from gvl_amd.config import make_opt                # noqa: E402
from gvl_amd.parallel import GraphedTrainStep      # noqa: E402
from gvl_amd.pdvc import build                     # noqa: E402
from gvl_amd.tuning import enable_tuned_gemms      # noqa: E402

enable_tuned_gemms()
dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
batches = rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
for policy in ("grow", "bucket"):
    tr = GraphedTrainStep(model, criterion, opt, world_size=1, cap_len_policy=policy)
    for _ in range(2):
        for dt in batches:
            tr(dt)
    torch.cuda.synchronize()

    def timed(seq):
        out = []
        for i in seq:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr(batches[i])
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) * 1e3)
        return out
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(40):
        tr(batches[k % 8])
    torch.cuda.synchronize()
    print(policy, f"un-synced loop of 40 rotating steps: {(time.perf_counter() - t0) * 1e3 / 40:.2f} ms per step")
    rot = timed(list(range(8)) * 3)[8:]
    same = {i: timed([i] * 6)[2:] for i in (0, 3, 5)}
    widths = [int(b["cap_tensor"].shape[-1]) for b in batches]
    print(policy, "graphs", len(tr.graphs), "caption tensor widths", widths)
    print("  rotating, per batch (ms):", " ".join(f"{sum(rot[i::8]) / len(rot[i::8]):.2f}" for i in range(8)))
    for i, v in same.items():
        print(f"  batch {i} repeated: {sum(v) / len(v):.2f} ms")
