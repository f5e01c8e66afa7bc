"""Dev tool: aten-op census (torch profiler) of eager training steps at bench shapes."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rotating_batches
from gvl_amd.config import make_opt
from gvl_amd.parallel import TrainStep
from gvl_amd.pdvc import build
from gvl_amd.tuning import enable_tuned_gemms

enable_tuned_gemms()
dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
batches = rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
step = TrainStep(model, criterion, opt, world_size=1)
for i in range(3):
    step(batches[i])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(4):
        step(batches[i])
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.self_device_time_total > 0 and not e.key.startswith(("Cijk", "void ", "(anonymous", "__amd"))]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print(f"# per step: {tot / 4 / 1e3:.2f} ms of device time in {sum(e.count for e in rows) / 4:.0f} op calls")
for e in rows[:40]:
    print(f"{e.key[:60]:60s} {e.count / 4:7.1f} calls {e.self_device_time_total / 4:8.1f} us")

print("# GEMMs by shape")
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm")]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:24]:
    print(f"{e.key:12s} {str(e.input_shapes)[:70]:70s} {e.count / 4:6.1f} calls {e.self_device_time_total / 4:8.1f} us  ({e.self_device_time_total / e.count:6.1f} each)")
