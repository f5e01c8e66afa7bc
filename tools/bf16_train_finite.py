#!/usr/bin/env python3
"""Dev tool (round 4): does the bf16 (autocast) train step of the bench stay finite?  Runs the eager TrainStep on the bench's
rotating batches and prints the loss and the first non-finite gradient / parameter per step."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rotating_batches                                           # noqa: E402
from gvl_amd.config import make_opt                                          # noqa: E402
from gvl_amd.pdvc import build                                               # noqa: E402
from gvl_amd.parallel import TrainStep                                       # noqa: E402
from gvl_amd.tuning import enable_tuned_gemms                                # noqa: E402

enable_tuned_gemms()
dev = torch.device("cuda:0")
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
batches = rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
tr = TrainStep(model, criterion, opt, autocast_dtype=torch.bfloat16 if dtype == "bf16" else None)
print("lr", opt.lr, "grad_clip", opt.grad_clip)
from gvl_amd import MultiScaleDeformableAttention as MSDA                  # noqa: E402
for step in range(int(os.environ.get("STEPS", 40))):
    MSDA.profile_enable(True)
    total, loss = tr(batches[step % 8])
    torch.cuda.synchronize()
    MSDA.profile_enable(False)
    kt = [f"{e[0]}[{e[1]}]={e[3]:.0f}" for e in MSDA.profile_collect() if "bwd" in str(e[0])]
    print("   ", " ".join(kt))
    bad_g = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    bad_p = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    off = [float(l.self_attn.sampling_offsets.weight.detach().abs().max()) for l in model.transformer.encoder.layers]
    off += [float(l.cross_attn.sampling_offsets.weight.detach().abs().max()) for l in model.transformer.decoder.layers]
    print(f"step {step:3d} loss {float(total):12.4f}  non-finite grads {len(bad_g)} {bad_g[:3]}  params {len(bad_p)} {bad_p[:2]}  "
          f"max|enc sampling_offsets.weight| {off}", flush=True)
    if bad_p:
        break
