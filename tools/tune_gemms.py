#!/usr/bin/env python3
"""(Re)generate gvl_amd/tunableop_mi355x.csv on an MI355X: runs the eval forward and the train step at the BASELINE
shapes (cfg A, and cfg L for the encoder/decoder GEMMs) with PyTorch TunableOp tuning enabled and writes the table.
    python tools/tune_gemms.py [out.csv]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_batch
from gvl_amd.config import make_opt
from gvl_amd.pdvc import build
from gvl_amd.parallel import TrainStep

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gvl_amd", "tunableop_mi355x.csv")
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(150)
torch.cuda.tunable.set_max_tuning_iterations(30)
torch.cuda.tunable.set_filename(out, False)      # TunableOp writes its table when the process exits
dev = torch.device("cuda:0")
for T in (100, 512):
    opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=T, device="cuda")
    torch.manual_seed(0)
    model, criterion, _, _ = build(opt)
    model = model.to(dev)
    dt = synth_batch(16, T, opt.feature_dim, opt.vocab_size, 3, dev)
    model.eval()
    with torch.no_grad():
        for _ in range(2):
            model(dt, criterion, None, "queries", eval_mode=True)
    model.train()
    tr = TrainStep(model, criterion, opt)
    for _ in range(2):
        tr(dt)
    torch.cuda.synchronize()
    print("tuned T =", T, "entries so far:", len(torch.cuda.tunable.get_results()))
print("table will be written on exit to", out)
