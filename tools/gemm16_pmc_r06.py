"""the two dominant launches of the inference token step at cfg A's shapes, five each, for rocprofv3 --pmc passes
(tools/gemm16_pmc_r06.sh): the vocabulary product with its argmax / sum-exp epilogue (k_vocab_f16x3: 4800 x 512 x 8518) and the
gate product with the LSTM cell as its epilogue (k_gates_f16x3: n = 4800, H = 512, K = 512 + 512)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvl_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)           # noqa: E731
n, H, Ka, V = 4800, 512, 512, 8518
xp = MSDA.split_rows(torch.tanh(rnd(n, H)))
wp = MSDA.split_rows(rnd(V, H) * 0.05)
b = rnd(V)
att, h_prev = rnd(n, Ka), torch.tanh(rnd(n, H))
w_cat = MSDA.split_rows((rnd(4 * H, H + Ka) * 0.04).contiguous())
gates_c, emb, c = rnd(n, 4 * H), rnd(V + 1, 4 * H), rnd(n, H)
it = torch.randint(0, V + 1, (n,), device=dev, generator=g)
ap_, hp = MSDA.split_rows(att), MSDA.split_rows(h_prev)
for _ in range(5):
    MSDA.gemm_f16x3_argmax(xp, wp, b)
    MSDA.gemm_f16x3_gates(ap_, hp, w_cat, gates_c, emb, it, c)
torch.cuda.synchronize()
