"""Is a gradient that disagrees with the reference ill-conditioned (sample positions crossing frame boundaries) or wrong?
Perturb the input features by 1e-6 / 1e-5 relative and watch the same gradient entries."""
import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import numpy as np, torch
from helpers import load
import test_gpu_full_dims as T
g = load("pdvc_anet_full_train")
names = ["transformer.reference_points.bias", "transformer.decoder.layers.1.cross_attn.sampling_offsets.bias",
         "transformer.decoder.layers.0.cross_attn.attention_weights.bias", "class_head.1.weight"]
f, opt, model, criterion = T.build_anet(True, transformer_dropout_prob=0.0, drop_prob=0.0)
base = {}
for scale in (1.0, 1.0 + 1e-6, 1.0 - 1e-6, 1.0 + 1e-5, 1.0 + 1e-4):
    dt = T.train_batch(f, g)
    dt["video_tensor"] = dt["video_tensor"] * scale
    model.zero_grad(set_to_none=True)
    out, loss = model(dt, criterion, None, "queries")
    wd = criterion.weight_dict
    final = sum(loss[k] * wd[k] for k in loss.keys() if k in wd)
    final.backward()
    P = dict(model.named_parameters())
    line = [f"scale-1={scale-1:+.0e} loss={float(final):.6f}"]
    for n in names:
        gr = P[n].grad.detach().double().cpu().numpy().ravel()
        if scale == 1.0:
            base[n] = gr
        ref = g["grad_norms"][[str(x) for x in g["grad_names"]].index(n)]
        line.append(f"{n.split('.')[-3] if n.count('.')>2 else n}: norm {np.linalg.norm(gr):.5f} (ref {float(ref):.5f}) d_base {np.abs(gr-base[n]).max():.2e}")
    print(" | ".join(line))
