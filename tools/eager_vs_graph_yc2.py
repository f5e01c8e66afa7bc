import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np, torch
import gvl_amd
from helpers import load, pdvc_state, pdvc_dt
from gvl_amd.config import make_opt
from gvl_amd.pdvc import build
from gvl_amd.parallel import GraphedEvalForward
dev = torch.device("cuda:0")
f = load("pdvc_yc2")
opt = make_opt("yc2_tsn_dvc", num_queries=int(f["num_queries"]), frame_embedding_num=512, device="cuda")
model, criterion, _, _ = build(opt)
model.load_state_dict(pdvc_state(f, seed=512), strict=True)
model = model.to(dev).eval()
dt = pdvc_dt(f, feat=int(f["feature_dim"]), seed=4)
dt = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in dt.items()}
dt["video_target"] = [{k: v.to(dev) for k, v in t_.items()} for t_ in dt["video_target"]]
with torch.no_grad():
    a, _ = model(dt, criterion, None, "queries", eval_mode=True)
    b, _ = model(dt, criterion, None, "queries", eval_mode=True)
g = GraphedEvalForward(model, criterion)
c, _ = g(dt)
c = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
d_, _ = g(dt)
def cmp(x, y, name):
    tok = float((x["seq"] == y["seq"]).float().mean()); cap = float((x["seq"] == y["seq"]).all(-1).float().mean())
    print(name, "tokens equal", tok, "captions equal", cap, "max d logits", float((x["pred_logits"] - y["pred_logits"]).abs().max()),
          "max d boxes", float((x["pred_boxes"] - y["pred_boxes"]).abs().max()))
cmp(a, b, "eager vs eager")
cmp(c, d_, "graph vs graph")
cmp(a, c, "eager vs graph")
ref = torch.from_numpy(f["seq"]).to(dev)
print("eager vs golden tokens", float((a["seq"][..., :ref.shape[-1]] == ref).float().mean()), "graph vs golden", float((c["seq"][..., :ref.shape[-1]] == ref).float().mean()))
