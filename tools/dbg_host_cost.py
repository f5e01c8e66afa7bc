"""Where do the ~800 us per call of the unfused decoder-shaped forward go (profiles/r01_kernel_sweep.txt)?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gvl_amd import MultiScaleDeformableAttention as MSDA, _lib
dev = torch.device("cuda:0")
B, M, D, L, P, T = 16, 8, 64, 4, 4, 100
lens = [100, 50, 25, 13]; S = sum(lens)
shapes = torch.tensor([(1, x) for x in lens], dtype=torch.long, device=dev)
lsi = torch.tensor([0, 100, 150, 175], dtype=torch.long, device=dev)
value = torch.randn(B, S, M, D, device=dev)
for Q in (188, 299, 300, 301, 320):
    loc = torch.rand(B, Q, M, L, P, 2, device=dev) * 1.5 - 0.25
    aw = torch.softmax(torch.randn(B, Q, M, L * P, device=dev), -1).view(B, Q, M, L, P)
    for _ in range(10):
        MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)
    t_issue = (time.perf_counter() - t0) / 50 * 1e6
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 50 * 1e6
    t0 = time.perf_counter()
    for _ in range(50):
        o = value.new_empty((B, Q, M * D))
    t_alloc = (time.perf_counter() - t0) / 50 * 1e6
    print(f"Q={Q}: host issue {t_issue:.1f} us/call, incl. drain {t_all:.1f} us/call, new_empty alone {t_alloc:.1f} us, out bytes {B*Q*M*D*4/1e6:.2f} MB")
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["segment.all.allocated"])
