#!/usr/bin/env python3
"""Dev tool (round 4): the bench's bf16 train half step by step -- GraphedTrainStep(autocast bf16, caption-width buckets) on the
rotating batches: loss per replay, non-finite parameters, and the deformable-attention backward's time in an eager
instrumented step every few replays (the r03 / r04 'bf16 backward is 2x slower in the bench' observation)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rotating_batches                                           # noqa: E402
from gvl_amd.config import make_opt                                          # noqa: E402
from gvl_amd.pdvc import build                                               # noqa: E402
from gvl_amd.parallel import GraphedTrainStep, TrainStep                     # noqa: E402
from gvl_amd.tuning import enable_tuned_gemms                                # noqa: E402
from gvl_amd import MultiScaleDeformableAttention as MSDA                    # noqa: E402

enable_tuned_gemms()
dev = torch.device("cuda:0")
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
batches = rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
ac = torch.bfloat16 if dtype == "bf16" else None
tr = GraphedTrainStep(model, criterion, opt, autocast_dtype=ac, cap_len_policy=os.environ.get("POLICY", "bucket"))
for dt in batches:
    tr(dt)


def probe(tag):
    MSDA.profile_enable(True)
    TrainStep.__call__(tr, batches[0])
    torch.cuda.synchronize()
    MSDA.profile_enable(False)
    kt = [f"{e[0]}[{e[1]}]={e[3]:.0f}" for e in MSDA.profile_collect() if "bwd_t1d" in str(e[0])]
    bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    offs = [float(l.cross_attn.sampling_offsets.weight.detach().abs().max()) for l in model.transformer.decoder.layers]
    offb = [float(l.cross_attn.sampling_offsets.bias.detach().abs().max()) for l in model.transformer.decoder.layers]
    print(f"  [{tag}] eager instrumented step: {' '.join(kt)} | non-finite params {len(bad)} {bad[:2]} | max|dec offsets W| {offs} |b| {offb}", flush=True)


for step in range(int(os.environ.get("STEPS", 24))):
    total, loss = tr(batches[step % 8])
    if os.environ.get("NOSYNC") and step < int(os.environ.get("STEPS", 24)) - 1:
        continue                                  # run ahead of the GPU like bench.py's timed loop: no host read per step
    fin = bool(torch.isfinite(total))
    if not fin or step % 8 == 7 or os.environ.get("NOSYNC"):
        bad = {k: float(v) for k, v in loss.items() if isinstance(v, torch.Tensor) and v.numel() == 1 and not torch.isfinite(v).all()}
        print(f"replay {step:3d} loss {float(total):10.4f} finite {fin} non-finite terms {bad}", flush=True)
    if not fin:
        gb = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        print("   non-finite grads:", len(gb), gb[:4])
        probe(f"after replay {step}")
        break
