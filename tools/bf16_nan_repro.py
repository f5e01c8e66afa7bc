#!/usr/bin/env python3
"""Dev tool (round 4): the bench's exact call order of the bf16 train half (set-up 0..7, warm-up 0..2, timed 0, 1, 2 -> NaN at the
14th call) through the EAGER TrainStep and through GraphedTrainStep: which of them goes non-finite, in which loss term and
which gradient first."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rotating_batches                                           # noqa: E402
from gvl_amd.config import make_opt                                          # noqa: E402
from gvl_amd.pdvc import build                                               # noqa: E402
from gvl_amd.parallel import GraphedTrainStep, TrainStep                     # noqa: E402
from gvl_amd.tuning import enable_tuned_gemms                                # noqa: E402

enable_tuned_gemms()
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "graphed"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
batches = rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
ac = torch.bfloat16 if dtype == "bf16" else None
tr = (GraphedTrainStep(model, criterion, opt, autocast_dtype=ac, cap_len_policy="bucket") if mode == "graphed"
      else TrainStep(model, criterion, opt, autocast_dtype=ac))
order = list(range(8)) + [0, 1, 2] + [i % 8 for i in range(10)]
for n, b in enumerate(order):
    total, loss = tr(batches[b])
    fin = bool(torch.isfinite(total))
    print(f"call {n:2d} batch {b} loss {float(total):9.4f}", flush=True)
    if not fin:
        bad = {k: float(v) for k, v in loss.items() if isinstance(v, torch.Tensor) and v.numel() == 1 and not torch.isfinite(v).all() and "self_iou" not in k}
        print("   non-finite loss terms (besides self_iou):", bad)
        gb = [nm for nm, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        print("   non-finite grads:", len(gb), gb[:6])
        pb = [nm for nm, p in model.named_parameters() if not torch.isfinite(p).all()]
        print("   non-finite params:", len(pb), pb[:3])
        break
