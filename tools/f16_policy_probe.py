#!/usr/bin/env python3
"""Dev tool (round 4): inference under torch.autocast with the three policies of gvl_amd.pdvc.autocast_inference_policy at
the bench shapes (cfg A) and the yc2 long-video shape: outputs against the plain fp32 forward, graph-replay time."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gvl_amd                                                              # noqa: E402,F401
from bench import rotating_batches                                          # noqa: E402
from gvl_amd.config import make_opt                                         # noqa: E402
from gvl_amd.parallel import GraphedEvalForward                             # noqa: E402
from gvl_amd.pdvc import build                                              # noqa: E402
from gvl_amd.tuning import enable_tuned_gemms                               # noqa: E402

enable_tuned_gemms()
dev = torch.device("cuda:0")
for cfg, T, Q in (("anet_tsp_ssvg", 100, 300), ("yc2_tsn_dvc", 512, 100)):
    opt = make_opt(cfg, num_queries=Q, frame_embedding_num=T, device="cuda")
    torch.manual_seed(0)
    model, criterion, _, _ = build(opt)
    model = model.to(dev).eval()
    batches = rotating_batches(4, 16, T, opt.feature_dim, opt.vocab_size, dev, seed=1)
    with torch.no_grad():
        ref, ref_loss = model(batches[0], criterion, None, "queries", eval_mode=True)
    for policy in ("fp32", "f16", "bf16"):
        os.environ["GVL_AUTOCAST_INFERENCE"] = "" if policy == "f16" else policy
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            out, loss = model(batches[0], criterion, None, "queries", eval_mode=True)
        d = {k: float((out[k].float() - ref[k].float()).abs().max()) for k in ("pred_boxes", "pred_logits", "pred_count")}
        n = min(out["seq"].shape[-1], ref["seq"].shape[-1])
        same = float((out["seq"][..., :n] == ref["seq"][..., :n]).float().mean())
        dl = {k: abs(float(loss[k]) - float(ref_loss[k])) for k in ("loss_ce", "loss_giou", "loss_counter")}
        g = GraphedEvalForward(model, criterion, autocast_dtype=torch.bfloat16)
        for b in batches:
            g(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            g(batches[i % 4])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        print(f"{cfg} T={T} policy {policy:5s}: {ms:6.2f} ms/step = {16 / ms * 1e3:7.1f} videos/s | vs fp32: "
              + " ".join(f"{k} {v:.2e}" for k, v in d.items()) + f" tokens equal {same:.4f} | "
              + " ".join(f"{k} {v:.2e}" for k, v in dl.items()), flush=True)
    os.environ.pop("GVL_AUTOCAST_INFERENCE", None)
    g0 = GraphedEvalForward(model, criterion)
    for b in batches:
        g0(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        g0(batches[i % 4])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{cfg} T={T} no autocast   : {ms:6.2f} ms/step = {16 / ms * 1e3:7.1f} videos/s", flush=True)
