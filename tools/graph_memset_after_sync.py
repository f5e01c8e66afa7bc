#!/usr/bin/env python3
"""Dev (round 4): does a captured torch reduction (its semaphore buffer is zero-filled with hipMemsetAsync per launch) still
replay correctly after torch.cuda.synchronize()?  Symptom in the bf16 train graph: bias gradients (column sums of bf16
matrices) come out as garbage in the second replay after a device synchronisation."""
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float32):
    for R, C in ((4800, 1536), (4800, 512), (3008, 2048), (76800, 512)):
        x = torch.randn(R, C, device=dev).to(dt)
        ref = x.float().sum(0)
        out = torch.empty(C, device=dev, dtype=dt)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                out.copy_(x.sum(0))
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out.copy_(x.sum(0))
        res = []
        for phase in ("before sync", "after sync"):
            if phase == "after sync":
                torch.cuda.synchronize()
            for k in range(4):
                out.fill_(float("nan"))
                # something else runs between the replays, as in a train loop
                tmp = torch.randn(1 << 20, device=dev).sum()
                g.replay()
                err = float((out.float() - ref).abs().max() / ref.abs().max())
                res.append(f"{err:.1e}")
        print(f"{str(dt)[6:]:8s} sum(0) of ({R}, {C}): relative error per replay, 4 before / 4 after torch.cuda.synchronize(): {' '.join(res)}", flush=True)
