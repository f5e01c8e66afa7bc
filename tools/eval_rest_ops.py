"""Dev tool: aten-op census (torch profiler) of one eager eval forward WITHOUT the captioner: where the ~3 ms outside the
token loop go."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rotating_batches
from gvl_amd.config import make_opt
from gvl_amd.pdvc import build
from gvl_amd.tuning import enable_tuned_gemms

enable_tuned_gemms()
dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, eval_disable_captioning=True, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).eval()
batches = rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
with torch.no_grad():
    for i in range(3):
        model(batches[i], criterion, None, "queries", eval_mode=True)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for i in range(4):
            model(batches[i], criterion, None, "queries", eval_mode=True)
        torch.cuda.synchronize()
import collections
by = collections.defaultdict(lambda: [0, 0.0])
for e in prof.key_averages(group_by_stack_n=8):
    if e.key in ("aten::copy_", "aten::cat", "aten::add", "aten::clamp_min", "aten::clamp", "aten::fill_", "aten::amax", "aten::mul", "aten::div"):
        frames = [f for f in e.stack if "/gvl_amd/" in f or "/bench.py" in f]
        where = frames[0].split("/gvl_amd/")[-1] if frames else (e.stack[0] if e.stack else "?")
        by[(e.key, where[:90])][0] += e.count
        by[(e.key, where[:90])][1] += e.self_device_time_total
for (k, w), (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k:16s} {n / 4:5.1f} calls {us / 4:7.1f} us  {w}")
