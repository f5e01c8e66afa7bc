"""Dev probe: LayerNorm -> Linear product pairs as the decoder runs them (k_ln_rows writes y, k_lin_f16x3 stages y), timed from a
captured graph; GVL_ROWS_XCD=1 makes k_ln_rows write each contiguous eighth of the rows from the XCD that stages it next."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import layers as L   # noqa: E402

dev = "cuda:0"


def timeit(fn, n=40, reps=5):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / n)
    return best


print("GVL_ROWS_XCD =", os.environ.get("GVL_ROWS_XCD"))
torch.manual_seed(0)
for R, K, N in [(4800, 512, 512), (4800, 512, 2048), (4800, 512, 1536), (3008, 512, 2048), (3008, 512, 512)]:
    x = torch.randn(R, K, device=dev)
    norm = torch.nn.LayerNorm(K).to(dev)
    w, b = torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
    W = L.Weights([(w, b)])
    out = torch.empty(R, N, device=dev)
    big = torch.empty(64 << 20, device=dev)                 # 256 MB written between the pairs: nothing survives in a cache

    def pair():
        y, am, _ = L.layer_norm(x, norm)
        L.linear(y, W, [L.seg(0, out, am)])

    def pair_cold():
        big.fill_(1.0)
        pair()

    def only_fill():
        big.fill_(1.0)

    t_pair, t_cold, t_fill = timeit(pair), timeit(pair_cold, n=10), timeit(only_fill, n=10)
    print(f"{R} x {K} x {N}: LN + product {t_pair:.2f} us back to back, {t_cold - t_fill:.2f} us behind a 256 MB fill")
