"""Does an external event pair recorded inside a captured graph time the kernel between them?  (ROCm 7.2 / torch 2.10)
Outcome on MI355X: no -- `RuntimeError: External events are disallowed in rocm` at the first record() under capture, so
bench.py cannot time a kernel INSIDE the replayed graph with events; its roofline blocks stamp eager launches after the
timed region and the rocprofv3 trace of the replays is committed beside them (profiles/r02_msda_launches_in_graph.txt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA
from gvl_amd.deformable_transformer import make_level_tensors
from gvl_amd.ops.modules.ms_deform_attn import temporal_shapes_2d

dev = torch.device("cuda:0")
tsh, lsi = make_level_tensors([100, 50, 25, 13], dev)
sh2 = temporal_shapes_2d(tsh, lsi)
value = torch.randn(16, 188, 8, 64, device=dev)
proj = torch.randn(16, 300, 256, device=dev)
ref = torch.rand(16, 300, 4, 2, device=dev) * 0.5
a = torch.randn(4800, 512, device=dev)
w = torch.randn(512, 512, device=dev)
e0 = torch.cuda.Event(enable_timing=True, external=True)
e1 = torch.cuda.Event(enable_timing=True, external=True)
e2 = torch.cuda.Event(enable_timing=True, external=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        (a @ w); MSDA.msda1d_fused_forward(value, sh2, lsi, proj, ref, 4, 4)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    b = a @ w
    e0.record()
    o = MSDA.msda1d_fused_forward(value, sh2, lsi, proj, ref, 4, 4)
    e1.record()
    c = b @ w
    e2.record()
for i in range(8):
    g.replay()
    torch.cuda.synchronize()
    print(f"replay {i}: msda interval {e0.elapsed_time(e1) * 1e3:7.2f} us | following GEMM interval {e1.elapsed_time(e2) * 1e3:7.2f} us")
MSDA.profile_enable(1)
for _ in range(5):
    (a @ w); MSDA.msda1d_fused_forward(value, sh2, lsi, proj, ref, 4, 4)
torch.cuda.synchronize()
MSDA.profile_enable(False)
print("eager stamped:", [round(us, 2) for t, a_, b_, us in MSDA.profile_collect() if t == "fwd_t1d_d64"])
