"""Dev tool: replay the captured eval forward on the same batch many times and compare every output bitwise (a race in the
LDS-DMA ring of the token-loop GEMMs would show up as a flipped token or a changed log-probability)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rotating_batches
from gvl_amd.config import make_opt
from gvl_amd.parallel import GraphedEvalForward
from gvl_amd.pdvc import build
from gvl_amd.tuning import enable_tuned_gemms

enable_tuned_gemms()
dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).eval()
batches = rotating_batches(4, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
g = GraphedEvalForward(model, criterion)
ref = {}
n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for rep in range(n_rep):
    for bi, dt in enumerate(batches):
        out, loss = g(dt)
        torch.cuda.synchronize()
        snap = {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
        snap.update({"loss_" + k: v.clone() for k, v in loss.items() if torch.is_tensor(v)})
        if bi not in ref:
            ref[bi] = snap
            continue
        for k, v in snap.items():
            r = ref[bi][k]
            if v.shape != r.shape or not torch.equal(v, r):
                bad += 1
                print(f"rep {rep} batch {bi}: {k} differs (max |d| {float((v.float() - r.float()).abs().max()) if v.shape == r.shape else 'shape'})")
print(f"{n_rep} x {len(batches)} replays, {len(ref[0])} tensors each: {bad} mismatches")
