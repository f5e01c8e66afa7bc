#!/usr/bin/env python3
"""(Re)generate gvl_amd/tunableop_mi355x.csv on an MI355X: runs the eval forward and the train step at the BASELINE
shapes (cfg A, and cfg L for the encoder/decoder GEMMs) with PyTorch TunableOp tuning enabled and writes the table.
    python tools/tune_gemms.py [out.csv]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_batch, rotating_batches
from gvl_amd.config import make_opt
from gvl_amd.pdvc import build
from gvl_amd.parallel import TrainStep, _PaddedBatch, _Capacity

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gvl_amd", "tunableop_mi355x.csv")
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(150)
torch.cuda.tunable.set_max_tuning_iterations(30)
if os.path.exists(out) and "--fresh" not in sys.argv:
    torch.cuda.tunable.read_file(out)            # keep what is already tuned, add the missing shapes
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
# the layout-independent (padded) train step at the steady-state capacities of bench.py's rotating workload: the
# teacher-forced captioner's GEMMs run on the compact fixed-capacity row set (2 x 96 rows x 23 steps at the default seed)
opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).train()
batches = rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
cap = _Capacity()
for dt in batches:
    cap.fit(dt, with_captions=True)
tr = TrainStep(model, criterion, opt)
pb = _PaddedBatch(batches[0], cap.slots, cap.cap_len, cap.pair_rows)
for dt in batches[:2]:
    pb.load(dt)
    tr(pb.dt)
torch.cuda.synchronize()
print("tuned padded train step at capacity", (cap.slots, cap.cap_len, cap.pair_rows), "entries:", len(torch.cuda.tunable.get_results()))
print("table will be written on exit to", out)
