#!/usr/bin/env python3
"""Dev: the bench's bf16 train half rebuilt piece by piece to find what makes its graphed step go NaN at timed step 2."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                 # noqa: E402
from gvl_amd.config import make_opt                                          # noqa: E402
from gvl_amd.pdvc import build                                               # noqa: E402
from gvl_amd.parallel import GraphedTrainStep                                # noqa: E402
from gvl_amd.tuning import enable_tuned_gemms                                # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "timed_loop"
os.environ["GVL_BENCH_TRACE_LOSS"] = "1"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
enable_tuned_gemms()
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, eval_disable_captioning=False, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev)
batches = bench.rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
model.train()
tr = GraphedTrainStep(model, criterion, opt, world_size=1, split_exchange=None, autocast_dtype=torch.bfloat16, cap_len_policy="bucket")
for i, dt in enumerate(batches):
    o = tr(dt)
    print(f"set-up {i} loss {float(o[0]):.4f}", file=sys.stderr)
if variant == "timed_loop":
    bench.timed_loop(tr, batches, 4, 3, 1, dev)
else:
    for n, i in enumerate([0, 1, 2, 0, 1, 2, 3]):
        if n == 3 and "sync" in variant:
            torch.cuda.synchronize()
        if n == 3 and "probe" in variant:
            from gvl_amd import MultiScaleDeformableAttention as _M
            print("clock", _M.clock_probe_mhz(dev, 4000), file=sys.stderr)
        if n == 3 and "sleep" in variant:
            import time
            time.sleep(0.5)
        o = tr(batches[i])
        if "nofloat" not in variant or n >= 5:
            print(f"manual batch {i} loss {float(o[0]):.4f}", file=sys.stderr)
