#!/usr/bin/env python3
"""Dev: where inside the graphed bf16 step do the NaNs start after a torch.cuda.synchronize()?  The three-graph form (forward +
backward | encoder backward | clip + Adam) lets the flat gradient buffer be inspected between the replays."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                 # noqa: E402
from gvl_amd.config import make_opt                                          # noqa: E402
from gvl_amd.pdvc import build                                               # noqa: E402
from gvl_amd.parallel import GraphedTrainStep                                # noqa: E402
from gvl_amd.tuning import enable_tuned_gemms                                # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
enable_tuned_gemms()
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else None
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
batches = bench.rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
tr = GraphedTrainStep(model, criterion, opt, world_size=1, split_exchange=True, autocast_dtype=dtype, cap_len_policy="bucket")
names = [n for n, p in model.named_parameters() if p.requires_grad]
state = {"call": -1}


def check():
    flat = tr.buckets.flat
    bad = []
    for n, p in zip(names, tr.params):
        if p.grad is not None and not bool(torch.isfinite(p.grad).all()):
            bad.append((n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()))
    big = [(n, f"{float(p.grad.abs().nan_to_num(nan=0.0, posinf=3e38).max()):.2e}", int((p.grad.abs() > 1e6).sum()), p.grad.numel())
           for n, p in zip(names, tr.params) if p.grad is not None and float(p.grad.abs().nan_to_num(nan=0.0, posinf=3e38).max()) > 1e6]
    if big:
        print(f"   call {state['call']}: |grad| > 1e6 in: {big[:10]}", file=sys.stderr, flush=True)
    print(f"   call {state['call']}: gradients before clip + Adam: {len(bad)} non-finite {bad[:8]}; |flat| max "
          f"{float(flat.abs().nan_to_num(nan=-1.0).max()):.3e}", file=sys.stderr, flush=True)


tr.buckets.exchange_end = check
order = list(range(8)) + [0, 1, 2] + ["sync"] + [0, 1, 2, 3]
for it in order:
    if it == "sync":
        kind = os.environ.get("SYNC", "device")
        if kind == "device":
            torch.cuda.synchronize()
        elif kind == "stream":
            torch.cuda.current_stream().synchronize()
        elif kind == "event":
            e_ = torch.cuda.Event(); e_.record(); e_.synchronize()
        print(f"-- sync ({kind})", file=sys.stderr)
        continue
    state["call"] += 1
    o = tr(batches[it])
    print(f"call {state['call']} batch {it} loss {float(o[0]):.4f}", file=sys.stderr, flush=True)
