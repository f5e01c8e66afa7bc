#!/usr/bin/env python3
"""Dev (round 4): which node kinds does a captured bf16 / fp32 column sum contain (hipGraphDebugDotPrint)?"""
import re
import torch
dev = torch.device("cuda:0")
for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(4800, 1536, device=dev).to(dt)
    out = torch.empty(1536, device=dev, dtype=dt)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out.copy_(x.sum(0))
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    g.enable_debug_mode()
    with torch.cuda.graph(g, stream=s):
        out.copy_(x.sum(0))
    path = f"/tmp/graph_{str(dt)[6:]}.dot"
    g.debug_dump(path)
    txt = open(path).read()
    kinds = re.findall(r'label="([^"]*)"', txt)
    print(dt, len(kinds), "nodes:")
    for k in kinds:
        print("   ", k.replace("\\n", " | ")[:200])
