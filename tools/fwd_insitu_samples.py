"""Dev tool: per-launch durations of the decoder / encoder k_fwd_t1d_d64 launches over several instrumented eager eval
steps (the figure behind bench.py's `roofline` block), to see its spread on the box at hand."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rotating_batches
from gvl_amd import MultiScaleDeformableAttention as MSDA
from gvl_amd.config import make_opt
from gvl_amd.pdvc import build
from gvl_amd.tuning import enable_tuned_gemms

enable_tuned_gemms()
dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).eval()
for h in model.caption_head:
    h.graph_decode = False
batches = rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
with torch.no_grad():
    for i in range(2):
        model(batches[i], criterion, None, "queries", eval_mode=True)
    torch.cuda.synchronize()
    for level in (1, 1):
        MSDA.profile_enable(level)
        for i in range(6):
            model(batches[i % 8], criterion, None, "queries", eval_mode=True)
        torch.cuda.synchronize()
        MSDA.profile_enable(False)
        rows = [(t, a, us) for t, a, b, us in MSDA.profile_collect() if t == "fwd_t1d_d64"]
        print("decoder (Lq=300):", " ".join(f"{us:5.2f}" for t, a, us in rows if a == 300))
        print("encoder (Lq=188):", " ".join(f"{us:5.2f}" for t, a, us in rows if a == 188))

    # does the reading depend on what the allocator has been through?  (bench.py stamps its eager steps AFTER the graphed,
    # timed region)
    from gvl_amd.parallel import GraphedEvalForward

    def sample(tag):
        MSDA.profile_enable(1)
        for i in range(4):
            model(batches[i % 8], criterion, None, "queries", eval_mode=True)
        torch.cuda.synchronize()
        MSDA.profile_enable(False)
        rows = [(a, us) for t, a, b, us in MSDA.profile_collect() if t == "fwd_t1d_d64"]
        dec = sorted(us for a, us in rows if a == 300)
        print(f"{tag}: decoder median {dec[len(dec) // 2]:5.2f} min {dec[0]:5.2f} | reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB")

    sample("before any graph      ")
    g = GraphedEvalForward(model, criterion)
    for dt in batches:
        g(dt)
    for i in range(10):
        g(batches[i % 8])
    torch.cuda.synchronize()
    sample("graphs alive          ")
    g.graphs.clear()
    del g
    sample("graphs deleted        ")
    torch.cuda.empty_cache()
    sample("after empty_cache     ")

    # right after a long stretch of graph replays (hot chip) vs after a pause
    import time
    g = GraphedEvalForward(model, criterion)
    for dt in batches:
        g(dt)
    for rep in range(2):
        for i in range(40):
            g(batches[i % 8])
        torch.cuda.synchronize()
        sample(f"right after 40 replays ")
        time.sleep(1.0)
        sample(f"after a 1 s pause      ")
