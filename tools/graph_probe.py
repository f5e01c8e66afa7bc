#!/usr/bin/env python3
"""Dev probe: replay the encoder+decoder from a hipGraph (kernels back-to-back) so that rocprofv3 shows the MSDA kernel
time without host launch gaps.   rocprofv3 --kernel-trace --stats -- python tools/graph_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from gvl_amd.config import make_opt
from gvl_amd.pdvc import build

dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, device="cuda", eval_disable_captioning=True)
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).eval()
dt = synth_batch(16, 100, 512, opt.vocab_size, 3, dev)
with torch.no_grad():
    for _ in range(3):
        model(dt, None, None, "queries", eval_mode=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        model(dt, None, None, "queries", eval_mode=True)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = model(dt, None, None, "queries", eval_mode=True)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    print("graph replay of encoder+decoder+heads (no captioner): %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
    t0 = time.perf_counter()
    for _ in range(50):
        model(dt, None, None, "queries", eval_mode=True)
    torch.cuda.synchronize()
    print("eager: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
