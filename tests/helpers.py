"""Shared test helpers: golden loading and synthetic weights (no reference access at run time)."""
import ast
import os

import numpy as np
import torch

from synth import synth_array, synth_state_dict, level_lengths  # noqa: F401  (tests/golden on sys.path)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def module_state(prefix, C=512, M=8, L=4, P=4, qdim=None, seed=0):
    """synthetic MSDeformAttn(-Cap) weights exactly as tests/golden/make_golden.py:load_synth produced them"""
    qdim = qdim or C
    shapes = {
        f"{prefix}sampling_offsets.weight": (M * L * P, qdim), f"{prefix}sampling_offsets.bias": (M * L * P,),
        f"{prefix}attention_weights.weight": (M * L * P, qdim), f"{prefix}attention_weights.bias": (M * L * P,),
        f"{prefix}value_proj.weight": (C, C), f"{prefix}value_proj.bias": (C,),
        f"{prefix}output_proj.weight": (C, C), f"{prefix}output_proj.bias": (C,),
    }
    return {k: t(v) for k, v in synth_state_dict(shapes, seed).items()}


def pdvc_state(fix, seed=300):
    names = [str(x) for x in fix["param_names"]]
    shapes = {n: ast.literal_eval(str(s)) for n, s in zip(names, fix["param_shapes"])}
    sd = {k: t(v) for k, v in synth_state_dict(shapes, seed).items()}
    # The reference model ties parameters that appear under two names (pdvc.py:124-140): the decoder's refinement
    # heads ARE model.bbox_head, and share_caption_head makes caption_head.0 the same module as caption_head.1.
    # load_state_dict(strict) writes the later key last, so the tied tensors end up with these values:
    for k in list(sd):
        if k.startswith("bbox_head."):
            sd["transformer.decoder." + k] = sd[k]
        if k.startswith("caption_head.1."):
            sd["caption_head.0." + k[len("caption_head.1."):]] = sd[k]
    return sd


def pdvc_dt(fix, feat=64, seed=1):
    T = int(fix["meta_T"])
    valid = [int(v) for v in fix["valid"]]
    n_gt = [int(v) for v in fix["n_gt"]]
    B = len(valid)
    vt = t(synth_array("dt.video_tensor", (B, T, feat), seed))
    vmask = torch.zeros(B, T, dtype=torch.bool)
    for i, v in enumerate(valid):
        vmask[i, :v] = True
        vt[i, v:] = 0
    vlen = torch.tensor([[float(v), 60.0 + 30.0 * i, float(n)] for i, (v, n) in enumerate(zip(valid, n_gt))])
    targets = []
    for i, n in enumerate(n_gt):
        c = t(synth_array(f"dt.gt_c{i}", (n,), seed, 0.25, 0.75))
        l_ = t(synth_array(f"dt.gt_l{i}", (n,), seed, 0.1, 0.4))
        targets.append({"boxes": torch.stack([c, l_], -1), "labels": torch.zeros(n, dtype=torch.long)})
    return {"video_tensor": vt, "video_mask": vmask, "video_length": vlen,
            "cap_raw": [["x"] * n for n in n_gt], "video_target": targets}


def maxerr(a, b):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max()) if a.numel() else 0.0


def path_census(fn):
    """run fn() with the library's per-dispatch stamps on (eager launches only) -> (fn's result, Counter of the kernel tags that
    ran): lets a model-level test assert that the hand-written layers served the call, not a silent PyTorch formulation
    """
    import collections
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    MSDA.profile_enable(2)
    try:
        MSDA.profile_collect()
        res = fn()
        torch.cuda.synchronize()
        tags = collections.Counter(e[0] for e in MSDA.profile_collect())
    finally:
        MSDA.profile_enable(0)
    return res, tags
