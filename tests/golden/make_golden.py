#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the *imported* reference on CPU (build container only).

    cd /tmp && python /root/repo/tests/golden/make_golden.py           # writes next to this file

The reference (zjr2000/GVL @ /root/reference) is imported in place through tests/golden/_refimport.py; nothing
of it is copied.  Each fixture holds inputs (or the seeds that regenerate them through tests/golden/synth.py)
and the reference's outputs.  Two padding semantics are recorded for the sampling op:

  *_zeros   the CUDA op behind MSDeformAttnFunction (pdvc/ops/src/cuda/ms_deform_im2col_cuda.cuh:238-300): obtained
            from the reference's own ms_deform_attn_core_pytorch with its grid_sample call switched to
            padding_mode='zeros' for the duration of the call (the .cu sources cannot be compiled here);
  *_border  ms_deform_attn_core_pytorch exactly as shipped (func.py:61-62) = the CPU fallback and the captioner path.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

FULL = "--no-pdvc" not in sys.argv
_refimport.install(full=FULL)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import pdvc.ops.functions.ms_deform_attn_func as RF  # noqa: E402
import pdvc.ops.modules.ms_deform_attn as RM  # noqa: E402
from pdvc.ops.modules import MSDeformAttn, MSDeformAttnCap  # noqa: E402
from pdvc.matcher import HungarianMatcher  # noqa: E402
from synth import synth_array, synth_state_dict, level_lengths  # noqa: E402

torch.set_num_threads(4)
_orig_grid_sample = F.grid_sample
_orig_core = RF.ms_deform_attn_core_pytorch


def core_with_pad(pad):
    """the reference core with its grid_sample padding switched (zeros = CUDA-op semantics)."""
    def core(*a, **k):
        def gs(*ga, **gk):
            gk["padding_mode"] = pad
            return _orig_grid_sample(*ga, **gk)
        RF.F.grid_sample = gs
        try:
            return _orig_core(*a, **k)
        finally:
            RF.F.grid_sample = _orig_grid_sample
    return core


class cuda_semantics:
    """Inside: MSDeformAttn's sampling core zero-pads (what MSDeformAttnFunction computes on a GPU), while
    MSDeformAttnCap keeps calling the shipped (border) core -- exactly the reference's behaviour on a CUDA device."""
    def __enter__(self):
        RM.ms_deform_attn_core_pytorch = core_with_pad("zeros")

    def __exit__(self, *a):
        RM.ms_deform_attn_core_pytorch = _orig_core


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


# ----------------------------------------------------------------------------------------------- op level
def op_case(name, shapes, B, M, D, Q, P, dtype, seed, loc_lo=-0.25, loc_hi=1.25, value_scale=1.0, normalise_w=False):
    shapes = torch.as_tensor(shapes, dtype=torch.long)
    L = shapes.shape[0]
    S = int(shapes.prod(1).sum())
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    g = torch.Generator().manual_seed(seed)
    value = (torch.randn(B, S, M, D, generator=g, dtype=torch.float64) * value_scale).to(dtype)
    loc = (torch.rand(B, Q, M, L, P, 2, generator=g, dtype=torch.float64) * (loc_hi - loc_lo) + loc_lo).to(dtype)
    if int(shapes[:, 0].max()) == 1:
        loc[..., 1] = 0.5                                          # ms_deform_attn.py:115-116
    if normalise_w:                                                # pdvc/ops/test.py:33-35 style
        aw = torch.rand(B, Q, M, L, P, generator=g, dtype=torch.float64) + 1e-5
        aw = (aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)).to(dtype)
    else:
        aw = torch.softmax(torch.randn(B, Q, M, L * P, generator=g, dtype=torch.float64), -1).view(B, Q, M, L, P).to(dtype)
    gout = torch.randn(B, Q, M * D, generator=g, dtype=torch.float64).to(dtype)
    rec = dict(value=value, shapes=shapes, lsi=lsi, loc=loc, aw=aw, gout=gout)
    for pad in ("zeros", "border"):
        v, l_, a = (t.clone().requires_grad_() for t in (value, loc, aw))
        out = core_with_pad(pad)(v, shapes, l_, a)
        out.backward(gout)
        rec.update({f"out_{pad}": out, f"gvalue_{pad}": v.grad, f"gloc_{pad}": l_.grad, f"gaw_{pad}": a.grad})
    rec["sample_border"] = _orig_core(value, shapes, loc, aw, return_value=True)
    save(name, **rec)


def make_op():
    t1d = [(1, t) for t in level_lengths(20)]                      # [20,10,5,3]
    op_case("op_t1d_d64_f32", t1d, B=2, M=2, D=64, Q=6, P=4, dtype=torch.float32, seed=11)
    op_case("op_t1d_d8_f64", t1d, B=2, M=3, D=8, Q=5, P=4, dtype=torch.float64, seed=12)
    op_case("op_t1d_cap_d512_f32", t1d, B=1, M=1, D=512, Q=3, P=4, dtype=torch.float32, seed=13)
    # the reference's own test geometry (pdvc/ops/test.py:21-28): 2-D levels (6,4),(3,2), N=1,M=2,Lq=2,L=2,P=2
    ref2d = [(6, 4), (3, 2)]
    for D in (2, 30, 32, 71):
        op_case(f"op_test2d_d{D}_f64", ref2d, B=1, M=2, D=D, Q=2, P=2, dtype=torch.float64, seed=3 + D,
                loc_lo=0.0, loc_hi=1.0, value_scale=0.01, normalise_w=True)
    op_case("op_test2d_d64_f32", ref2d, B=1, M=2, D=64, Q=2, P=2, dtype=torch.float32, seed=3,
            loc_lo=0.0, loc_hi=1.0, value_scale=0.01, normalise_w=True)
    op_case("op_2d_edges_f32", [(5, 7), (3, 4), (1, 2)], B=2, M=2, D=16, Q=9, P=3, dtype=torch.float32, seed=21,
            loc_lo=-0.4, loc_hi=1.4)


# ------------------------------------------------------------------------------------------- module level
def load_synth(module, seed, prefix=""):
    shapes = {prefix + k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth_state_dict(shapes, seed)
    module.load_state_dict({k[len(prefix):]: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return sd


def make_module():
    C, M, L, P = 512, 8, 4, 4
    T = level_lengths(12)
    S = sum(T)
    tshapes = torch.as_tensor(T, dtype=torch.long)
    lsi = torch.cat((tshapes.new_zeros(1), tshapes.cumsum(0)[:-1]))
    B, Q = 2, 6
    for refdim in (1, 2):
        torch.manual_seed(0)
        mod = MSDeformAttn(C, L, M, P).eval()
        load_synth(mod, seed=100 + refdim, prefix="attn.")
        query = torch.from_numpy(synth_array(f"mod{refdim}.query", (B, Q, C), 1))
        inp = torch.from_numpy(synth_array(f"mod{refdim}.input", (B, S, C), 1))
        ref = torch.from_numpy(synth_array(f"mod{refdim}.ref", (B, Q, L, refdim), 1, 0.05, 0.95))
        if refdim == 2:
            ref[..., 1] = ref[..., 1] * 0.5
        mask = torch.zeros(B, S, dtype=torch.bool)
        for l in range(L):                                          # video 1: last third of every level is padding
            mask[1, int(lsi[l]) + (2 * T[l]) // 3: int(lsi[l]) + T[l]] = True
        gout = torch.from_numpy(synth_array(f"mod{refdim}.gout", (B, Q, C), 1))
        rec = dict(tshapes=tshapes, lsi=lsi, mask=mask, meta=np.array([B, Q, C, M, L, P, refdim]))
        for pad in ("zeros", "border"):
            RM.ms_deform_attn_core_pytorch = core_with_pad(pad)
            try:
                q, x = query.clone().requires_grad_(), inp.clone().requires_grad_()
                mod.zero_grad()
                out = mod(q, ref, x, tshapes, lsi, mask)
                out.backward(gout)
                rec.update({f"out_{pad}": out, f"gquery_{pad}": q.grad, f"ginput_{pad}": x.grad,
                            f"g_off_w_{pad}": mod.sampling_offsets.weight.grad,
                            f"g_aw_b_{pad}": mod.attention_weights.bias.grad,
                            f"g_vproj_b_{pad}": mod.value_proj.bias.grad})
            finally:
                RM.ms_deform_attn_core_pytorch = _orig_core
        save(f"module_ref{refdim}", **rec)

    # captioner variant: query dim 2C, heads 1, returns unweighted samples
    torch.manual_seed(0)
    Qc = 3
    cap = MSDeformAttnCap(C, L, 1, P).eval()
    load_synth(cap, seed=200, prefix="cap.")
    query = torch.from_numpy(synth_array("cap.query", (B, Qc, 2 * C), 1))
    inp = torch.from_numpy(synth_array("cap.input", (B, S, C), 1))
    ref = torch.from_numpy(synth_array("cap.ref", (B, Qc, L, 2), 1, 0.05, 0.95))
    ref[..., 1] *= 0.5
    mask = torch.zeros(B, S, dtype=torch.bool)
    mask[1, S - 2:] = True
    out = cap(query, ref, inp, tshapes, lsi, mask)
    save("module_cap", tshapes=tshapes, lsi=lsi, mask=mask, out=out, meta=np.array([B, Qc, C, 1, L, P, 2]))


def make_module_cap3c():
    """MSDeformAttnCap with enable_pos_emb_for_captioner (3C-wide queries: ms_deform_attn_for_caption.py:54-56), the shipped
    (border) core as the reference calls it, same inputs as module_cap otherwise"""
    import types
    B, T, C, L, P = 2, 37, 64, 4, 4
    lens = level_lengths(T, L)
    S = sum(lens)
    tshapes = torch.tensor(lens, dtype=torch.long)
    lsi = torch.cat([tshapes.new_zeros(1), tshapes.cumsum(0)[:-1]])
    torch.manual_seed(0)
    Qc = 5
    cap = MSDeformAttnCap(C, L, 1, P, opt=types.SimpleNamespace(enable_pos_emb_for_captioner=True)).eval()
    assert cap.sampling_offsets.in_features == 3 * C
    load_synth(cap, seed=210, prefix="cap3.")
    query = torch.from_numpy(synth_array("cap3.query", (B, Qc, 3 * C), 1))
    inp = torch.from_numpy(synth_array("cap3.input", (B, S, C), 1))
    ref = torch.from_numpy(synth_array("cap3.ref", (B, Qc, L, 2), 1, 0.05, 0.95))
    ref[..., 1] *= 0.5
    mask = torch.zeros(B, S, dtype=torch.bool)
    mask[1, S - 3:] = True
    out = cap(query, ref, inp, tshapes, lsi, mask)
    save("module_cap3c", tshapes=tshapes, lsi=lsi, mask=mask, out=out, meta=np.array([B, Qc, C, 1, L, P, 2]))


# ---------------------------------------------------------------------------------------------- matcher
def make_matcher():
    rec = {}
    m = HungarianMatcher(cost_class=2, cost_bbox=0, cost_giou=4, cost_alpha=0.25, cost_gamma=2, cost_cl=2.0)
    g = torch.Generator().manual_seed(5)
    B, Q = 3, 12
    sizes = [3, 1, 5]
    logits = torch.randn(B, Q, 1, generator=g)
    boxes = torch.rand(B, Q, 2, generator=g) * 0.5 + 0.2
    targets = [{"labels": torch.zeros(n, dtype=torch.long),
                "boxes": torch.rand(n, 2, generator=g) * 0.5 + 0.25} for n in sizes]
    indices, rl, C = m({"pred_logits": logits, "pred_boxes": boxes, "cl_match_mats": 0}, targets, return_C=True)
    rec.update(logits=logits, boxes=boxes, sizes=np.array(sizes), tgt_boxes=torch.cat([t["boxes"] for t in targets]))
    for i in range(B):
        rec[f"C_{i}"] = C[i]
        rec[f"idx_{i}"] = torch.stack(indices[i])
        rec[f"rl_{i}"] = torch.stack(rl[i])
    save("matcher_model", **rec)

    # raw LSAP contract on explicit cost matrices incl. ties / near ties, float32 -> scipy
    from scipy.optimize import linear_sum_assignment
    rs = np.random.RandomState(7)
    mats = {
        "ones_5x2": np.ones((5, 2), np.float32),
        "tie_3x2": np.array([[1, 1], [1, 1], [0, 1]], np.float32),
        "rand_300x3": rs.rand(300, 3).astype(np.float32),
        "rand_300x30": rs.rand(300, 30).astype(np.float32),
        "rand_30x30": rs.rand(30, 30).astype(np.float32),
        "rand_7x19": rs.rand(7, 19).astype(np.float32),
        "quant_40x12": np.round(rs.rand(40, 12) * 4).astype(np.float32),          # many exact ties
        "quant_12x40": np.round(rs.rand(12, 40) * 3).astype(np.float32),
        "tiled_50x4x4": np.tile(rs.rand(50, 4).astype(np.float32), (1, 4)),       # the m2o tiling, matcher.py:125-126
        "neg_20x6": (rs.rand(20, 6) - 0.5).astype(np.float32) * 10,
        "const_rows_9x4": np.repeat(rs.rand(9, 1).astype(np.float32), 4, axis=1),
        "single_col_6x1": rs.rand(6, 1).astype(np.float32),
        "single_row_1x6": rs.rand(1, 6).astype(np.float32),
    }
    rec = {}
    for k, c in mats.items():
        r, cidx = linear_sum_assignment(torch.from_numpy(c))          # same call form as matcher.py:124
        rec[f"{k}.C"] = c
        rec[f"{k}.rows"] = np.asarray(r, np.int64)
        rec[f"{k}.cols"] = np.asarray(cidx, np.int64)
    import scipy
    rec["scipy_version"] = np.array(scipy.__version__)
    save("lsap_cases", **rec)


# ------------------------------------------------------------------------------------ PDVC / transformer
PDVC_OVERRIDES = dict(num_queries=8, feature_dim=64, vocab_size=40, max_caption_len=6, enable_contrastive=False,
                      device="cpu")


def build_pdvc(cfg="cfgs/anet_tsp_ssvg.yml", overrides=None):
    import opts
    import pdvc.pdvc as P
    cwd = os.getcwd()
    os.makedirs("/tmp/gvl_golden_scratch", exist_ok=True)
    os.chdir("/tmp/gvl_golden_scratch")                              # parse_opts writes ./.tmp/opts.json
    argv = sys.argv
    sys.argv = ["x", "--cfg_path", os.path.join(_refimport.REF, cfg)]
    try:
        opt = opts.parse_opts()
    finally:
        sys.argv = argv
        os.chdir(cwd)
    for k, v in (PDVC_OVERRIDES if overrides is None else overrides).items():
        setattr(opt, k, v)
    torch.manual_seed(0)
    model, criterion, cc, post_ = P.build(opt)
    globals()["post"] = post_
    return opt, model.eval(), criterion, cc


post = None


def synth_dt(B, T, feat, valid, n_gt, seed=1):
    vt = torch.from_numpy(synth_array("dt.video_tensor", (B, T, feat), seed))
    vmask = torch.zeros(B, T, dtype=torch.bool)
    for i, v in enumerate(valid):
        vmask[i, :v] = True
        vt[i, v:] = 0
    vlen = torch.tensor([[float(v), 60.0 + 30.0 * i, float(n)] for i, (v, n) in enumerate(zip(valid, n_gt))])
    targets = []
    for i, n in enumerate(n_gt):
        c = torch.from_numpy(synth_array(f"dt.gt_c{i}", (n,), seed, 0.25, 0.75))
        l_ = torch.from_numpy(synth_array(f"dt.gt_l{i}", (n,), seed, 0.1, 0.4))
        targets.append({"boxes": torch.stack([c, l_], -1), "labels": torch.zeros(n, dtype=torch.long)})
    return {"video_tensor": vt, "video_mask": vmask, "video_length": vlen,
            "cap_raw": [["x"] * n for n in n_gt], "video_target": targets}


def make_pdvc():
    opt, model, criterion, cc = build_pdvc()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed=300)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    B, T = 2, 24
    dt = synth_dt(B, T, opt.feature_dim, valid=[24, 17], n_gt=[3, 2])
    rec = dict(meta_T=np.array(T), valid=np.array([24, 17]), n_gt=np.array([3, 2]),
               param_names=np.array(sorted(shapes)), param_shapes=np.array([str(shapes[k]) for k in sorted(shapes)]))
    for tag in ("cuda", "cpu"):
        ctx = cuda_semantics() if tag == "cuda" else None
        if ctx:
            ctx.__enter__()
        try:
            with torch.no_grad():
                # stage-wise, following PDVC.forward (pdvc.py:250-275)
                vf, mask, dur = dt["video_tensor"], ~dt["video_mask"], dt["video_length"][:, 1]
                srcs, masks, pos = model.base_encoder(vf, mask, dur)
                enc_in = model.transformer.prepare_encoder_inputs(srcs, masks, pos)
                src_flatten, tshapes, lsi, vr, lvl_pos, mflat = enc_in
                memory = model.transformer.forward_encoder(*enc_in)
                qe = model.query_embed.weight
                pmask = torch.ones(B, qe.shape[0]).bool()
                init_ref, tgt, refp, qpos = model.transformer.prepare_decoder_input_query(memory, qe)
                hs, inter = model.transformer.forward_decoder(tgt, refp, memory, tshapes, lsi, vr, qpos, mflat, pmask,
                                                              False)
                out, loss = model(dt, criterion, cc, "queries", eval_mode=True)
            rec.update({f"{tag}.src_flatten": src_flatten, f"{tag}.lvl_pos": lvl_pos, f"{tag}.valid_ratios": vr,
                        f"{tag}.mask_flatten": mflat, f"{tag}.memory": memory, f"{tag}.hs": hs,
                        f"{tag}.inter_references": inter, f"{tag}.init_reference": init_ref,
                        f"{tag}.pred_logits": out["pred_logits"], f"{tag}.pred_boxes": out["pred_boxes"],
                        f"{tag}.pred_count": out["pred_count"], f"{tag}.seq": out["seq"],
                        f"{tag}.cap_prob_eval": out["caption_probs"]["cap_prob_eval"],
                        f"{tag}.aux_pred_logits": out["aux_outputs"][0]["pred_logits"],
                        f"{tag}.aux_pred_boxes": out["aux_outputs"][0]["pred_boxes"]})
            if tag == "cuda":
                # PostProcess (pdvc.py:1003-1089) on these outputs, with a stand-in translator (token ids -> string)
                class _Tr:
                    @staticmethod
                    def rtranslate(s):
                        return " ".join(str(int(x)) for x in s if x > 0)
                loader = type("L", (), {"dataset": type("D", (), {"translator": _Tr})})
                res = post["bbox"](out, dt["video_length"][:, 1], loader)
                for i, r in enumerate(res):
                    rec[f"post.{i}.scores"] = r["scores"]
                    rec[f"post.{i}.labels"] = r["labels"]
                    rec[f"post.{i}.boxes"] = r["boxes"]
                    rec[f"post.{i}.query_id"] = r["query_id"]
                    rec[f"post.{i}.raw_idx"] = r["raw_idx"]
                    rec[f"post.{i}.pred_seq_len"] = r["pred_seq_len"]
                    rec[f"post.{i}.caption_scores"] = np.asarray(r["caption_scores"], np.float64)
                    rec[f"post.{i}.captions"] = np.array(r["captions"])
            for i, (a, b) in enumerate(out["matched_indices"][0]):
                rec[f"{tag}.match_{i}"] = torch.stack([a, b])
            for i, (a, b) in enumerate(out["matched_indices"][1]):
                rec[f"{tag}.rl_match_{i}"] = torch.stack([a, b])
            for k, v in loss.items():
                rec[f"{tag}.loss.{k}"] = torch.as_tensor(v)
        finally:
            if ctx:
                ctx.__exit__()
    rec["tshapes"], rec["lsi"] = tshapes, lsi
    save("pdvc_eval", **rec)

    # one captioner step in isolation (LSTM_DSA.py:120-124,241-271) on the cuda-semantics memory
    cap = model.caption_head[-1]
    with torch.no_grad():
        Q = qe.shape[0]
        hs_last = hs[-1]
        ref_in = inter[0][:, :, None] * torch.stack([vr] * 2, -1)[:, None]
        g = torch.Generator().manual_seed(9)
        h0 = torch.randn(1, B * Q, 512, generator=g) * 0.3
        c0 = torch.randn(1, B * Q, 512, generator=g) * 0.3
        it = torch.randint(0, opt.vocab_size + 1, (B * Q,), generator=g)
        logp, (h1, c1) = cap.get_logprobs_state(it, (h0, c0), hs_last, ref_in, memory, tshapes, lsi, mflat)
    save("captioner_step", it=it, h0=h0[0], c0=c0[0], hs=hs_last, ref_in=ref_in, memory=memory, mask=mflat,
         tshapes=tshapes, lsi=lsi, logp=logp, h1=h1[0], c1=c1[0])


def gt_proposal_inputs(dt):
    """dt['gt_boxes'] / dt['gt_boxes_mask'] as the reference's collate builds them (video_dataset.py: padded
    (B, max_gt, 2) boxes + bool mask)."""
    n = [len(tg["boxes"]) for tg in dt["video_target"]]
    boxes = torch.zeros(len(n), max(n), 2)
    mask = torch.zeros(len(n), max(n), dtype=torch.bool)
    for i, tg in enumerate(dt["video_target"]):
        boxes[i, :n[i]] = tg["boxes"]
        mask[i, :n[i]] = True
    return boxes, mask


def make_gtprop():
    """Evaluation with ground-truth proposals as decoder input (transformer_input_type='gt_proposals':
    misc/utils.py:32-43 decide_two_stage, deformable_transformer.py:137-147 prepare_decoder_input_proposal,
    iterative refinement disabled), CUDA-op semantics."""
    import copy
    opt, model, criterion, cc = build_pdvc()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed=300)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    B, T = 2, 24
    dt = synth_dt(B, T, opt.feature_dim, valid=[24, 17], n_gt=[3, 2])
    dt["gt_boxes"], dt["gt_boxes_mask"] = gt_proposal_inputs(dt)
    crit = copy.deepcopy(criterion)                       # decide_two_stage zeroes weights / the caption cost in place
    with cuda_semantics(), torch.no_grad():
        out, loss = model(dt, crit, cc, "gt_proposals", eval_mode=True)
    rec = dict(meta_T=np.array(T), valid=np.array([24, 17]), n_gt=np.array([3, 2]),
               param_names=np.array(sorted(shapes)), param_shapes=np.array([str(shapes[k]) for k in sorted(shapes)]),
               gt_boxes=dt["gt_boxes"], gt_boxes_mask=dt["gt_boxes_mask"],
               pred_logits=out["pred_logits"], pred_boxes=out["pred_boxes"], pred_count=out["pred_count"],
               seq=out["seq"], cap_prob_eval=out["caption_probs"]["cap_prob_eval"],
               aux_pred_boxes=out["aux_outputs"][0]["pred_boxes"],
               aux_pred_logits=out["aux_outputs"][0]["pred_logits"])
    for i, (a, b) in enumerate(out["matched_indices"][0]):
        rec[f"match_{i}"] = torch.stack([a, b])
    for k, v in loss.items():
        rec[f"loss.{k}"] = torch.as_tensor(v)
    rec["weight_names"] = np.array(sorted(crit.weight_dict))
    rec["weight_values"] = np.array([float(crit.weight_dict[k]) for k in sorted(crit.weight_dict)])
    save("pdvc_gtprop", **rec)


def make_yc2():
    """BASELINE.json config 4 (cfgs/yc2_tsn_dvc.yml: 3072-d TSN features, 100 queries, vocabulary 1607) on long
    videos, T = 512 -> levels 512/256/128/64, S = 960: evaluation forward of the reference at the config's REAL model
    dimensions (contrastive branch off: RoBERTa weights are not available offline; caption length capped at 8 to
    bound the CPU time).  CUDA-op semantics.  Pins the long-video (level 0 in global memory) kernels in situ."""
    opt, model, criterion, cc = build_pdvc("cfgs/yc2_tsn_dvc.yml",
                                           dict(enable_contrastive=False, device="cpu", max_caption_len=8,
                                                frame_embedding_num=512))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed=512)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    B, T = 2, 512
    valid, n_gt = [512, 389], [4, 3]
    dt = synth_dt(B, T, opt.feature_dim, valid=valid, n_gt=n_gt, seed=4)
    with cuda_semantics(), torch.no_grad():
        vf, mask, dur = dt["video_tensor"], ~dt["video_mask"], dt["video_length"][:, 1]
        srcs, masks, pos = model.base_encoder(vf, mask, dur)
        enc_in = model.transformer.prepare_encoder_inputs(srcs, masks, pos)
        memory = model.transformer.forward_encoder(*enc_in)
        out, loss = model(dt, criterion, cc, "queries", eval_mode=True)
    rec = dict(meta_T=np.array(T), valid=np.array(valid), n_gt=np.array(n_gt), feature_dim=np.array(opt.feature_dim),
               num_queries=np.array(opt.num_queries), vocab_size=np.array(opt.vocab_size),
               param_names=np.array(sorted(shapes)), param_shapes=np.array([str(shapes[k]) for k in sorted(shapes)]),
               tshapes=enc_in[1], lsi=enc_in[2],
               # memory (2, 960, 512) is the largest tensor: keep every 8th row + its exact sum as the pin
               memory_rows=memory[:, ::8], memory_sum=memory.double().sum(),
               pred_logits=out["pred_logits"], pred_boxes=out["pred_boxes"], pred_count=out["pred_count"],
               seq=out["seq"], cap_prob_eval=out["caption_probs"]["cap_prob_eval"],
               aux_pred_boxes=out["aux_outputs"][0]["pred_boxes"], event_feat=out["event_feat"][:, ::4])
    for i, (a, b) in enumerate(out["matched_indices"][0]):
        rec[f"match_{i}"] = torch.stack([a, b])
    for k, v in loss.items():
        rec[f"loss.{k}"] = torch.as_tensor(v)
    save("pdvc_yc2", **rec)


def make_yc2_train():
    """One TRAINING forward / backward of the reference on the long-video configuration of make_yc2 (cfgs/yc2_tsn_dvc.yml, T = 512
    -> S = 960, 100 queries, vocabulary 1607), B = 8 (B x heads = 64: the smallest batch the row-ownership backward serves),
    every dropout 0, captions of 3..6 words, CUDA-op semantics: every loss term,
    the matcher indices, the float64-accumulated gradient norm of EVERY parameter.  Pins the long-video BACKWARD kernels (level 0
    in global memory, rows owned across query chunks) inside the real model."""
    opt, model, criterion, cc = build_pdvc("cfgs/yc2_tsn_dvc.yml",
                                           dict(enable_contrastive=False, device="cpu", max_caption_len=8,
                                                frame_embedding_num=512, transformer_dropout_prob=0.0, drop_prob=0.0))
    model.train()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed=512)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    B, T = 8, 512
    valid, n_gt = [512, 389, 512, 277, 450, 512, 130, 498], [4, 3, 0, 2, 6, 1, 2, 5]
    dt = synth_dt(B, T, opt.feature_dim, valid=valid, n_gt=n_gt, seed=4)
    g = torch.Generator().manual_seed(23)
    cap_len = 8
    words = torch.randint(3, cap_len - 1, (sum(n_gt),), generator=g)              # words per caption: 3..6
    caps = torch.zeros(sum(n_gt), cap_len, dtype=torch.long)
    cap_mask = torch.zeros(sum(n_gt), cap_len)
    for i, w in enumerate(words.tolist()):
        caps[i, 1:1 + w] = torch.randint(1, opt.vocab_size, (w,), generator=g)
        cap_mask[i, :w + 2] = 1
    mx = max(n_gt)
    dt.update(cap_tensor=caps, cap_mask=cap_mask,
              gt_boxes_mask=torch.tensor([[k < n for k in range(mx)] for n in n_gt]).bool(),
              gt_gather_idx=torch.tensor([i for i, n in enumerate(n_gt) for _ in range(n)]))
    with cuda_semantics():
        out, loss = model(dt, criterion, cc, "queries")
        wd = criterion.weight_dict
        final = sum(loss[k] * wd[k] for k in loss.keys() if k in wd)
        final.backward()
    rec = dict(meta_T=np.array(T), valid=np.array(valid), n_gt=np.array(n_gt), feature_dim=np.array(opt.feature_dim),
               num_queries=np.array(opt.num_queries), vocab_size=np.array(opt.vocab_size),
               cap_tensor=caps, cap_mask=cap_mask, final_loss=final.detach(),
               param_names=np.array(sorted(shapes)), param_shapes=np.array([str(shapes[k]) for k in sorted(shapes)]))
    for k, v in loss.items():
        rec[f"loss.{k}"] = torch.as_tensor(v).detach()
    names = sorted(n for n, p_ in model.named_parameters() if p_.grad is not None)
    params = dict(model.named_parameters())
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = torch.stack([params[n].grad.double().norm() for n in names])
    for n in ("transformer.encoder.layers.0.self_attn.value_proj.weight", "transformer.decoder.layers.1.cross_attn.value_proj.bias",
              "transformer.level_embed", "class_head.1.weight", "count_head.0.bias"):
        rec["grad." + n] = params[n].grad if params[n].grad.numel() <= 4096 else params[n].grad[::16, ::8]
    for i, (a, b) in enumerate(out["matched_indices"][0]):
        rec[f"match_{i}"] = torch.stack([a, b])
    # the same step in float64: the reference's OWN fp32 error of every gradient (at T = 512 the sampling-location gradients --
    # differences of neighbouring frames times T_l -- make every parameter upstream of an offsets projection noisy)
    grads32 = {n: params[n].grad.clone() for n in names}
    model.zero_grad(set_to_none=True)
    model.double()
    dt64 = dict(dt)
    dt64["video_tensor"], dt64["video_length"] = dt["video_tensor"].double(), dt["video_length"].double()
    dt64["cap_mask"] = cap_mask.double()
    dt64["video_target"] = [{"boxes": t_["boxes"].double(), "labels": t_["labels"]} for t_ in dt["video_target"]]
    torch.set_default_dtype(torch.float64)
    try:
        with cuda_semantics():
            out64, loss64 = model(dt64, criterion, cc, "queries")
            final64 = sum(loss64[k] * wd[k] for k in loss64.keys() if k in wd)
            final64.backward()
    finally:
        torch.set_default_dtype(torch.float32)
    rec["final_loss_f64"] = final64.detach()
    rec["grad_norms_f64"] = torch.stack([params[n].grad.norm() for n in names])
    rec["grad_norm_f32_err"] = torch.stack([(grads32[n].double() - params[n].grad).norm() for n in names])
    rec["match_same_in_f64"] = np.array(all(torch.equal(torch.stack(list(a)), torch.stack(list(b)))
                                            for a, b in zip(out["matched_indices"][0], out64["matched_indices"][0])))
    save("pdvc_yc2_train", **rec)


def make_anet_full():
    """BASELINE.json configs 1-2 at the REAL model dimensions (cfgs/anet_tsp_ssvg.yml: 512-d TSP features, T = 100,
    300 queries, vocabulary 8517, 30 caption tokens) on a padded 2-video batch: evaluation forward of the reference,
    CUDA-op semantics."""
    opt, model, criterion, cc = build_pdvc("cfgs/anet_tsp_ssvg.yml",
                                           dict(enable_contrastive=False, device="cpu", num_queries=300,
                                                frame_embedding_num=100))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed=100)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    B, T = 2, 100
    valid, n_gt = [100, 73], [3, 5]
    dt = synth_dt(B, T, opt.feature_dim, valid=valid, n_gt=n_gt, seed=6)
    with cuda_semantics(), torch.no_grad():
        vf, mask, dur = dt["video_tensor"], ~dt["video_mask"], dt["video_length"][:, 1]
        srcs, masks, pos = model.base_encoder(vf, mask, dur)
        enc_in = model.transformer.prepare_encoder_inputs(srcs, masks, pos)
        memory = model.transformer.forward_encoder(*enc_in)
        out, loss = model(dt, criterion, cc, "queries", eval_mode=True)
    rec = dict(meta_T=np.array(T), valid=np.array(valid), n_gt=np.array(n_gt), feature_dim=np.array(opt.feature_dim),
               num_queries=np.array(opt.num_queries), vocab_size=np.array(opt.vocab_size),
               max_caption_len=np.array(opt.max_caption_len),
               param_names=np.array(sorted(shapes)), param_shapes=np.array([str(shapes[k]) for k in sorted(shapes)]),
               memory_rows=memory[:, ::4], memory_sum=memory.double().sum(),
               pred_logits=out["pred_logits"], pred_boxes=out["pred_boxes"], pred_count=out["pred_count"],
               seq=out["seq"], cap_prob_eval=out["caption_probs"]["cap_prob_eval"],
               aux_pred_boxes=out["aux_outputs"][0]["pred_boxes"], event_feat=out["event_feat"][:, ::8])
    for i, (a, b) in enumerate(out["matched_indices"][0]):
        rec[f"match_{i}"] = torch.stack([a, b])
    for k, v in loss.items():
        rec[f"loss.{k}"] = torch.as_tensor(v)
    save("pdvc_anet_full", **rec)


def make_anet_full_b16():
    """The HEADLINE workload's batch (BASELINE.json config 1: cfgs/anet_tsp_ssvg.yml, B = 16, T = 100, 300 queries,
    vocabulary 8517, 30 caption tokens) through the reference's evaluation forward, CUDA-op semantics: 16 videos of
    different valid lengths (incl. full and short ones) with 0..10 events each -- VERDICT r3 weak 1(a): the B = 2 fixture
    above left the 16-video batch checked at op level only.  Compact record: heads, counts, refined boxes of both decoder
    layers, greedy tokens of every query, log-probabilities of every 4th query, matched indices, losses."""
    opt, model, criterion, cc = build_pdvc("cfgs/anet_tsp_ssvg.yml",
                                           dict(enable_contrastive=False, device="cpu", num_queries=300,
                                                frame_embedding_num=100))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed=100)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    B, T = 16, 100
    valid = [100, 73, 100, 41, 88, 100, 57, 96, 100, 100, 64, 29, 100, 81, 100, 50]
    n_gt = [3, 5, 0, 1, 10, 2, 4, 7, 1, 6, 3, 2, 8, 0, 5, 4]
    dt = synth_dt(B, T, opt.feature_dim, valid=valid, n_gt=n_gt, seed=16)
    with cuda_semantics(), torch.no_grad():
        out, loss = model(dt, criterion, cc, "queries", eval_mode=True)
    rec = dict(meta_T=np.array(T), valid=np.array(valid), n_gt=np.array(n_gt), feature_dim=np.array(opt.feature_dim),
               num_queries=np.array(opt.num_queries), vocab_size=np.array(opt.vocab_size),
               max_caption_len=np.array(opt.max_caption_len),
               param_names=np.array(sorted(shapes)), param_shapes=np.array([str(shapes[k]) for k in sorted(shapes)]),
               pred_logits=out["pred_logits"], pred_boxes=out["pred_boxes"], pred_count=out["pred_count"],
               seq=out["seq"].to(torch.int16), cap_prob_eval=out["caption_probs"]["cap_prob_eval"][:, ::4],
               aux_pred_boxes=out["aux_outputs"][0]["pred_boxes"], event_feat=out["event_feat"][:, ::8, ::4])
    for i, (a, b) in enumerate(out["matched_indices"][0]):
        rec[f"match_{i}"] = torch.stack([a, b])
    for k, v in loss.items():
        rec[f"loss.{k}"] = torch.as_tensor(v)
    save("pdvc_anet_full_b16", **rec)


def make_full_train_probe():
    """VERDICT r3 missing 3: what does the reference do for set_cost_caption > 0 (pdvc.py:305-309 ->
    parallel_prediction_full_train, :322-432)?  With the LSTM-DSA captioner ('standard', every cfg of the path) its
    caption_prediction is called with indices=None and fails at pdvc.py:743; the exception type, message and the raising
    line are recorded (there is no output to pin)."""
    import traceback
    opt, model, criterion, cc = build_pdvc("cfgs/anet_tsp_ssvg.yml",
                                           dict(enable_contrastive=False, device="cpu", num_queries=20,
                                                frame_embedding_num=40, set_cost_caption=1.0,
                                                transformer_dropout_prob=0.0, drop_prob=0.0))
    model.train()
    n_gt = [2, 3]
    dt = synth_dt(2, 40, opt.feature_dim, valid=[40, 31], n_gt=n_gt, seed=6)
    g = torch.Generator().manual_seed(21)
    caps = torch.zeros(sum(n_gt), 8, dtype=torch.long)
    cap_mask = torch.zeros(sum(n_gt), 8)
    for i in range(sum(n_gt)):
        caps[i, 1:5] = torch.randint(1, opt.vocab_size, (4,), generator=g)
        cap_mask[i, :6] = 1
    dt.update(cap_tensor=caps, cap_mask=cap_mask,
              gt_boxes_mask=torch.tensor([[k < n for k in range(max(n_gt))] for n in n_gt]).bool(),
              gt_gather_idx=torch.tensor([i for i, n in enumerate(n_gt) for _ in range(n)]))
    rec = dict(raised=np.array(False), exc_type=np.array(""), exc_message=np.array(""), where=np.array(""))
    try:
        with cuda_semantics():
            model(dt, criterion, cc, "queries")
    except Exception as e:                                   # noqa: BLE001
        tb = traceback.extract_tb(e.__traceback__)[-1]
        rec = dict(raised=np.array(True), exc_type=np.array(type(e).__name__), exc_message=np.array(str(e)),
                   where=np.array(f"{os.path.basename(tb.filename)}:{tb.lineno} in {tb.name}"))
    save("full_train_probe", **rec)


def make_collate():
    """the reference's collate_fn (video_dataset.py:16-106) on synthetic samples (tests/golden/synth.py:synth_samples)"""
    from itertools import chain as chain_
    from synth import synth_samples
    import video_dataset as VD
    dt = VD.collate_fn(synth_samples())
    rec = {k: v for k, v in dt.items() if isinstance(v, torch.Tensor)}
    rec["video_key"] = np.array(dt["video_key"])
    rec["gt_featstamps"] = np.array(dt["gt_featstamps"])
    rec["cap_raw"] = np.array(list(chain_(*dt["cap_raw"])))
    for i, tg in enumerate(dt["video_target"]):
        rec[f"target.{i}.boxes"], rec[f"target.{i}.labels"] = tg["boxes"], tg["labels"]
    rec["keys"] = np.array(sorted(dt))
    save("collate", **rec)


def make_dataset():
    """the reference's PropSeqDataset.__getitem__ (video_dataset.py:209-281) + collate_fn on the synthetic on-disk
    dataset of tests/golden/synth.py:synth_dataset, both feature-type branches of load_feats"""
    import tempfile
    from synth import synth_dataset, dataset_opt
    import video_dataset as VD
    root = tempfile.mkdtemp(prefix="gvl_ds_")
    info = synth_dataset(root)
    rec = {}
    for kind in ("tsp", "c3d"):
        opt = dataset_opt(kind, info["vocab_size"])
        folder = [info["tsp_dir"]] if kind == "tsp" else info["c3d_dir"]
        ds = VD.PropSeqDataset(info["anno"], folder, info["vocab"], True, "gt", opt)
        np.random.seed(7)
        samples = [ds[i] for i in range(len(ds))]
        rec[f"{kind}.n"] = np.array(len(ds))
        for i, (feats, featstamps, labels, caps, stamps, dur, raw, key) in enumerate(samples):
            pre = f"{kind}.{i}."
            rec[pre + "feats"] = np.asarray(feats)
            rec[pre + "featstamps"] = np.asarray(featstamps).reshape(-1, 2)
            rec[pre + "labels"] = np.asarray(labels)
            rec[pre + "cap_lens"] = np.array([len(c) for c in caps])
            rec[pre + "caps"] = np.concatenate(caps)
            rec[pre + "stamps"] = np.asarray(stamps, dtype=np.float64).reshape(-1, 2)
            rec[pre + "duration"] = np.array(dur)
            rec[pre + "raw"] = np.array(raw)
            rec[pre + "key"] = np.array(key)
        dt = VD.collate_fn(samples[:3])
        rec[f"{kind}.collate.video_tensor"] = dt["video_tensor"]
        rec[f"{kind}.collate.cap_tensor"] = dt["cap_tensor"]
        rec[f"{kind}.collate.video_length"] = dt["video_length"]
    tr = VD.Translator(info["vocab"], info["vocab_size"])
    rec["rtranslate"] = np.array([tr.rtranslate([3, 7, 0, 5]), tr.rtranslate([0, 1]), tr.rtranslate([2, 22, 4])])
    save("dataset", **rec)


def make_anet_c3d():
    """BASELINE.json config 0 -- cfgs/anet_c3d_ssvg.yml (500-d C3D features, the file's own 30 queries) at its real
    dimensions: evaluation forward of the reference on a padded 2-video batch, CUDA-op semantics."""
    opt, model, criterion, cc = build_pdvc("cfgs/anet_c3d_ssvg.yml", dict(enable_contrastive=False, device="cpu"))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed=500)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    B, T = 2, 100
    valid, n_gt = [100, 61], [2, 4]
    dt = synth_dt(B, T, opt.feature_dim, valid=valid, n_gt=n_gt, seed=8)
    with cuda_semantics(), torch.no_grad():
        out, loss = model(dt, criterion, cc, "queries", eval_mode=True)
    rec = dict(meta_T=np.array(T), valid=np.array(valid), n_gt=np.array(n_gt), feature_dim=np.array(opt.feature_dim),
               num_queries=np.array(opt.num_queries), vocab_size=np.array(opt.vocab_size),
               max_caption_len=np.array(opt.max_caption_len),
               param_names=np.array(sorted(shapes)), param_shapes=np.array([str(shapes[k]) for k in sorted(shapes)]),
               pred_logits=out["pred_logits"], pred_boxes=out["pred_boxes"], pred_count=out["pred_count"],
               seq=out["seq"], cap_prob_eval=out["caption_probs"]["cap_prob_eval"],
               aux_pred_boxes=out["aux_outputs"][0]["pred_boxes"])
    for i, (a, b) in enumerate(out["matched_indices"][0]):
        rec[f"match_{i}"] = torch.stack([a, b])
    for k, v in loss.items():
        rec[f"loss.{k}"] = torch.as_tensor(v)
    save("pdvc_anet_c3d", **rec)


def _f64_eval(cfg, overrides, seed_w, B, T, valid, n_gt, seed_dt):
    """the SAME eval forward in float64 (model.double(), double inputs): the value both fp32 implementations approximate"""
    opt, model, criterion, cc = build_pdvc(cfg, overrides)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed=seed_w)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    dt = synth_dt(B, T, opt.feature_dim, valid=valid, n_gt=n_gt, seed=seed_dt)
    with cuda_semantics(), torch.no_grad():
        out32, _ = model(dt, criterion, cc, "queries", eval_mode=True)
    model = model.double()
    criterion.counter_class_rate = criterion.counter_class_rate.double() if hasattr(criterion, "counter_class_rate") else None
    dt64 = dict(dt)
    dt64["video_tensor"] = dt["video_tensor"].double()
    dt64["video_length"] = dt["video_length"].double()
    dt64["video_target"] = [{"boxes": t_["boxes"].double(), "labels": t_["labels"]} for t_ in dt["video_target"]]
    torch.set_default_dtype(torch.float64)            # the reference creates several tensors with the default type
    try:
        with cuda_semantics(), torch.no_grad():
            out64, _ = model(dt64, criterion, cc, "queries", eval_mode=True)
    finally:
        torch.set_default_dtype(torch.float32)
    rec = {}
    for k in ("pred_logits", "pred_boxes", "pred_count"):
        rec[k + "_f64"] = out64[k].double()
        rec[k + "_f32_err"] = (out32[k].double() - out64[k].double()).abs().max()      # the REFERENCE's own fp32 error
    rec["cap_prob_eval_f32_err"] = (out32["caption_probs"]["cap_prob_eval"].double()
                                    - out64["caption_probs"]["cap_prob_eval"].double())[out32["seq"] == out64["seq"]].abs().max()
    rec["seq_agree"] = (out32["seq"] == out64["seq"]).double().mean()
    rec["seq_f64"] = out64["seq"]
    return rec


def make_f64():
    """fp64 evaluations of the two full-dimension eval fixtures (pdvc_anet_full, pdvc_yc2): the noise floor of an fp32
    evaluation of these models = the reference's OWN fp32 run against its fp64 run.  The GPU tests bound gvl_amd's error
    against the fp64 values by a small multiple of that floor instead of an unexplained tolerance."""
    rec = _f64_eval("cfgs/anet_tsp_ssvg.yml", dict(enable_contrastive=False, device="cpu", num_queries=300,
                                                   frame_embedding_num=100), 100, 2, 100, [100, 73], [3, 5], 6)
    save("pdvc_anet_full_f64", **rec)


def make_anet_full_train():
    """BASELINE.json config 2 at the REAL dimensions: one training forward/backward of the reference on
    cfgs/anet_tsp_ssvg.yml (300 queries, vocabulary 8517, T = 100), B = 2, every dropout 0, captions of up to 12 tokens:
    every loss term, the matcher indices of both decoder layers, the gradient norm of EVERY parameter and a few
    gradients element-wise (pdvc.py:540-620, train.py:403-406).  CUDA-op semantics."""
    opt, model, criterion, cc = build_pdvc("cfgs/anet_tsp_ssvg.yml",
                                           dict(enable_contrastive=False, device="cpu", num_queries=300,
                                                frame_embedding_num=100, transformer_dropout_prob=0.0, drop_prob=0.0))
    model.train()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed=100)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    B, T = 2, 100
    valid, n_gt = [100, 73], [3, 5]
    dt = synth_dt(B, T, opt.feature_dim, valid=valid, n_gt=n_gt, seed=6)
    g = torch.Generator().manual_seed(21)
    cap_len = 12
    words = torch.randint(3, cap_len - 1, (sum(n_gt),), generator=g)              # words per caption: 3..10
    caps = torch.zeros(sum(n_gt), cap_len, dtype=torch.long)
    cap_mask = torch.zeros(sum(n_gt), cap_len)
    for i, w in enumerate(words.tolist()):
        caps[i, 1:1 + w] = torch.randint(1, opt.vocab_size, (w,), generator=g)
        cap_mask[i, :w + 2] = 1
    mx = max(n_gt)
    dt.update(cap_tensor=caps, cap_mask=cap_mask,
              gt_boxes_mask=torch.tensor([[k < n for k in range(mx)] for n in n_gt]).bool(),
              gt_gather_idx=torch.tensor([i for i, n in enumerate(n_gt) for _ in range(n)]))
    # the operands and gradients of ONE deformable-attention op inside this very step (decoder
    # layer 1, cross attention; video 1, the padded one), recorded at the op's own boundary -- the outputs of value_proj /
    # sampling_offsets / attention_weights, the reference points, the gradient arriving at the op's output and the
    # gradients leaving through its three inputs.  A test that feeds these to the fused HIP op isolates the kernel's
    # sampling-offset gradients from the summation order of the GEMMs upstream of it.
    att = model.transformer.decoder.layers[1].cross_attn
    cap = {}

    def keep(name):
        def hook(mod, inp, outp):
            outp.retain_grad()
            cap[name] = outp
        return hook

    def pre(mod, args):
        cap["ref"], cap["mask"] = args[1].detach(), args[5]

    def pre_out(mod, args):
        args[0].retain_grad()
        cap["op_out"] = args[0]
    hooks = [att.value_proj.register_forward_hook(keep("value")), att.sampling_offsets.register_forward_hook(keep("off")),
             att.attention_weights.register_forward_hook(keep("logit")), att.register_forward_pre_hook(pre),
             att.output_proj.register_forward_pre_hook(pre_out)]
    with cuda_semantics():
        out, loss = model(dt, criterion, cc, "queries")
        wd = criterion.weight_dict
        final = sum(loss[k] * wd[k] for k in loss.keys() if k in wd)
        final.backward()
    for h_ in hooks:
        h_.remove()
    vb = 1
    save("msda_op_in_train_step", video=np.array(vb), value=cap["value"][vb].detach(), off=cap["off"][vb].detach(),
         logit=cap["logit"][vb].detach(), ref=cap["ref"][vb], mask=cap["mask"][vb], out=cap["op_out"][vb].detach(),
         grad_out=cap["op_out"].grad[vb], grad_value=cap["value"].grad[vb], grad_off=cap["off"].grad[vb],
         grad_logit=cap["logit"].grad[vb], lens=np.array(level_lengths(T)))
    rec = dict(meta_T=np.array(T), valid=np.array(valid), n_gt=np.array(n_gt), cap_tensor=caps, cap_mask=cap_mask,
               final_loss=final.detach(),
               param_names=np.array(sorted(shapes)), param_shapes=np.array([str(shapes[k]) for k in sorted(shapes)]))
    for k, v in loss.items():
        rec[f"loss.{k}"] = torch.as_tensor(v).detach()
    names = sorted(n for n, p_ in model.named_parameters() if p_.grad is not None)
    params = dict(model.named_parameters())
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = torch.stack([params[n].grad.norm() for n in names])
    for n in ("transformer.encoder.layers.0.self_attn.sampling_offsets.bias",
              "transformer.decoder.layers.0.cross_attn.sampling_offsets.bias",
              "transformer.decoder.layers.1.cross_attn.attention_weights.bias",
              "transformer.decoder.layers.1.cross_attn.sampling_offsets.bias",
              "transformer.level_embed", "caption_head.0.core.deformable_att.sampling_offsets.bias",
              "caption_head.0.core.alpha_net.weight", "class_head.1.weight", "count_head.0.bias",
              "transformer.decoder.bbox_head.0.layers.2.bias", "transformer.reference_points.weight"):
        rec["grad." + n] = params[n].grad
    rec["grad_rows.caption_head.0.logit.weight"] = params["caption_head.0.logit.weight"].grad[::97]
    rec["grad_rows.query_embed.weight"] = params["query_embed.weight"].grad[::13]
    for i, (a, b) in enumerate(out["matched_indices"][0]):
        rec[f"match_{i}"] = torch.stack([a, b])
    # the same step in float64: the gradient of a sampling location is piecewise constant in the location (difference of
    # the two neighbouring frames), so an fp32 rounding that moves a sample across a frame boundary changes it by O(1);
    # the reference's own fp32-vs-fp64 deviation of every gradient norm is the noise floor the GPU tests bound against
    grads32 = {n: params[n].grad.clone() for n in names}
    model.zero_grad(set_to_none=True)
    model.double()
    dt64 = dict(dt)
    dt64["video_tensor"], dt64["video_length"] = dt["video_tensor"].double(), dt["video_length"].double()
    dt64["cap_mask"] = cap_mask.double()
    dt64["video_target"] = [{"boxes": t_["boxes"].double(), "labels": t_["labels"]} for t_ in dt["video_target"]]
    torch.set_default_dtype(torch.float64)
    try:
        with cuda_semantics():
            out64, loss64 = model(dt64, criterion, cc, "queries")
            final64 = sum(loss64[k] * wd[k] for k in loss64.keys() if k in wd)
            final64.backward()
    finally:
        torch.set_default_dtype(torch.float32)
    rec["final_loss_f64"] = final64.detach()
    rec["grad_f64.transformer.decoder.layers.1.cross_attn.sampling_offsets.bias"] = \
        params["transformer.decoder.layers.1.cross_attn.sampling_offsets.bias"].grad
    rec["grad_norms_f64"] = torch.stack([params[n].grad.norm() for n in names])
    rec["grad_norm_f32_err"] = torch.stack([(grads32[n].double() - params[n].grad).norm() for n in names])
    for k, v in loss64.items():
        rec[f"loss_f64.{k}"] = torch.as_tensor(v).detach()
    same = all(torch.equal(torch.stack(list(a)), torch.stack(list(b)))
               for a, b in zip(out["matched_indices"][0], out64["matched_indices"][0]))
    rec["match_same_in_f64"] = np.array(same)
    save("pdvc_anet_full_train", **rec)


def make_anet_full_train_b16():
    """The train step of make_anet_full_train at the HEADLINE batch: B = 16 videos of different valid lengths with 0..10 events
    each (the batch of make_anet_full_b16), captions of 3..10 words, every dropout 0, CUDA-op semantics: every loss term, the
    matcher indices of both decoder layers, the gradient norm of EVERY parameter (pdvc.py:540-620, train.py:403-406)."""
    opt, model, criterion, cc = build_pdvc("cfgs/anet_tsp_ssvg.yml",
                                           dict(enable_contrastive=False, device="cpu", num_queries=300,
                                                frame_embedding_num=100, transformer_dropout_prob=0.0, drop_prob=0.0))
    model.train()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed=100)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    B, T = 16, 100
    valid = [100, 73, 100, 41, 88, 100, 57, 96, 100, 100, 64, 29, 100, 81, 100, 50]
    n_gt = [3, 5, 0, 1, 10, 2, 4, 7, 1, 6, 3, 2, 8, 0, 5, 4]
    dt = synth_dt(B, T, opt.feature_dim, valid=valid, n_gt=n_gt, seed=16)
    g = torch.Generator().manual_seed(22)
    cap_len = 12
    words = torch.randint(3, cap_len - 1, (sum(n_gt),), generator=g)              # words per caption: 3..10
    caps = torch.zeros(sum(n_gt), cap_len, dtype=torch.long)
    cap_mask = torch.zeros(sum(n_gt), cap_len)
    for i, w in enumerate(words.tolist()):
        caps[i, 1:1 + w] = torch.randint(1, opt.vocab_size, (w,), generator=g)
        cap_mask[i, :w + 2] = 1
    mx = max(n_gt)
    dt.update(cap_tensor=caps, cap_mask=cap_mask,
              gt_boxes_mask=torch.tensor([[k < n for k in range(mx)] for n in n_gt]).bool(),
              gt_gather_idx=torch.tensor([i for i, n in enumerate(n_gt) for _ in range(n)]))
    with cuda_semantics():
        out, loss = model(dt, criterion, cc, "queries")
        wd = criterion.weight_dict
        final = sum(loss[k] * wd[k] for k in loss.keys() if k in wd)
        final.backward()
    rec = dict(meta_T=np.array(T), valid=np.array(valid), n_gt=np.array(n_gt), feature_dim=np.array(opt.feature_dim),
               cap_tensor=caps, cap_mask=cap_mask, final_loss=final.detach(),
               param_names=np.array(sorted(shapes)), param_shapes=np.array([str(shapes[k]) for k in sorted(shapes)]))
    for k, v in loss.items():
        rec[f"loss.{k}"] = torch.as_tensor(v).detach()
    names = sorted(n for n, p_ in model.named_parameters() if p_.grad is not None)
    params = dict(model.named_parameters())
    rec["grad_names"] = np.array(names)
    # norms accumulated in float64 (an fp32 accumulation over the 4.4 M elements of logit.weight is itself off by 1.5e-3)
    rec["grad_norms"] = torch.stack([params[n].grad.double().norm() for n in names])
    rec["grad_rows.caption_head.0.logit.weight"] = params["caption_head.0.logit.weight"].grad[::97]
    rec["grad.class_head.1.weight"] = params["class_head.1.weight"].grad
    rec["grad.count_head.0.bias"] = params["count_head.0.bias"].grad
    rec["grad.transformer.level_embed"] = params["transformer.level_embed"].grad
    for i, (a, b) in enumerate(out["matched_indices"][0]):
        rec[f"match_{i}"] = torch.stack([a, b])
    save("pdvc_anet_full_train_b16", **rec)


def make_train():
    """One training forward/backward of the reference (pdvc.py parallel_prediction_matched, train.py:403-406) with
    every dropout probability set to 0 so that the step is deterministic; CUDA-op (zero padding) semantics."""
    global PDVC_OVERRIDES
    saved = dict(PDVC_OVERRIDES)
    PDVC_OVERRIDES.update(transformer_dropout_prob=0.0, drop_prob=0.0)
    try:
        opt, model, criterion, cc = build_pdvc()
    finally:
        PDVC_OVERRIDES = saved
    model.train()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed=300)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    B, T = 2, 24
    n_gt = [3, 2]
    dt = synth_dt(B, T, opt.feature_dim, valid=[24, 17], n_gt=n_gt)
    g = torch.Generator().manual_seed(11)
    cap_len = 7
    caps = torch.randint(1, opt.vocab_size, (sum(n_gt), cap_len), generator=g)
    caps[:, 0] = 0
    caps[:, -1] = 0
    caps[1, 4:] = 0                                                   # a shorter caption
    cap_mask = (torch.arange(cap_len)[None] <= (caps != 0).sum(1)[:, None] + 0).float()
    dt.update(cap_tensor=caps, cap_mask=cap_mask, gt_boxes_mask=torch.tensor([[1, 1, 1], [1, 1, 0]]).bool(),
              gt_gather_idx=torch.tensor([0, 0, 0, 1, 1]))
    with cuda_semantics():
        out, loss = model(dt, criterion, cc, "queries")
        wd = criterion.weight_dict
        final = sum(loss[k] * wd[k] for k in loss.keys() if k in wd)
        final.backward()
    rec = dict(cap_tensor=caps, cap_mask=cap_mask, final_loss=final.detach())
    for k, v in loss.items():
        rec[f"loss.{k}"] = torch.as_tensor(v).detach()
    names = sorted(n for n, p_ in model.named_parameters() if p_.grad is not None)
    params = dict(model.named_parameters())
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = torch.stack([params[n].grad.norm() for n in names])
    for n in ("transformer.encoder.layers.0.self_attn.sampling_offsets.bias",
              "transformer.encoder.layers.1.self_attn.attention_weights.bias",
              "transformer.decoder.layers.0.cross_attn.sampling_offsets.bias",
              "transformer.decoder.layers.1.cross_attn.value_proj.bias",
              "transformer.level_embed", "caption_head.0.core.deformable_att.sampling_offsets.bias",
              "caption_head.0.core.alpha_net.weight", "base_encoder.input_proj.0.1.bias", "class_head.1.weight", "transformer.decoder.bbox_head.0.layers.2.bias"):
        rec["grad." + n] = params[n].grad
    for i, (a, b) in enumerate(out["matched_indices"][0]):
        rec[f"match_{i}"] = torch.stack([a, b])
    save("pdvc_train", **rec)


def make_switches(only=None):
    """The reference's configuration switches on the path that no other fixture exercises: the eval forward
    with eval_disable_captioning, with_box_refine = 0, share_caption_head = 0, and a TRAINING forward / backward with
    caption_loss_coef = 0 (pdvc.py:262-275 then routes training through parallel_prediction_full).  Small dimensions, CUDA-op
    semantics, the loaded state dict is what load_state_dict leaves (tied parameters: the later key wins)."""
    global PDVC_OVERRIDES
    B, T = 2, 24
    cases = {"nocap": dict(eval_disable_captioning=True), "norefine": dict(with_box_refine=0), "unshared": dict(share_caption_head=0),
             "nocaploss": dict(caption_loss_coef=0, transformer_dropout_prob=0.0, drop_prob=0.0),
             # round 6: the captioner sees [hs | query_embed] -- 3C-wide MSDeformAttnCap projections
             # (ms_deform_attn_for_caption.py:54-56), 3C LSTM input (LSTM_DSA.py), pdvc.py:344
             "posemb": dict(enable_pos_emb_for_captioner=True)}
    if only:
        cases = {k: v for k, v in cases.items() if k in only}
    for name, over in cases.items():
        saved = dict(PDVC_OVERRIDES)
        PDVC_OVERRIDES.update(over)
        try:
            opt, model, criterion, cc = build_pdvc()
        finally:
            PDVC_OVERRIDES = saved
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(shapes, seed=310).items()}, strict=True)
        loaded = model.state_dict()                                   # (tied parameters share their storage)
        dt = synth_dt(B, T, opt.feature_dim, valid=[24, 17], n_gt=[3, 2])
        rec = dict(meta_T=np.array(T), valid=np.array([24, 17]), n_gt=np.array([3, 2]), param_names=np.array(sorted(shapes)),
                   param_shapes=np.array([str(shapes[k]) for k in sorted(shapes)]))
        # which names ended up tied to which (the test rebuilds the loaded state from the synthetic one)
        ties = []
        names = sorted(shapes)
        for i, a in enumerate(names):
            for b in names[i + 1:]:
                if loaded[a].data_ptr() == loaded[b].data_ptr() and a != b:
                    ties.append(f"{a}={b}")
        rec["ties"] = np.array(ties if ties else ["-"])
        with cuda_semantics():
            if name == "nocaploss":
                model.train()
                g = torch.Generator().manual_seed(11)
                caps = torch.randint(1, opt.vocab_size, (5, 7), generator=g)
                caps[:, 0] = 0
                caps[:, -1] = 0
                cap_mask = (torch.arange(7)[None] <= (caps != 0).sum(1)[:, None]).float()
                dt.update(cap_tensor=caps, cap_mask=cap_mask, gt_boxes_mask=torch.tensor([[1, 1, 1], [1, 1, 0]]).bool(),
                          gt_gather_idx=torch.tensor([0, 0, 0, 1, 1]))
                out, loss = model(dt, criterion, cc, "queries")
                wd = criterion.weight_dict
                final = sum(loss[k] * wd[k] for k in loss.keys() if k in wd)
                final.backward()
                gn = sorted(n for n, p_ in model.named_parameters() if p_.grad is not None)
                params = dict(model.named_parameters())
                rec.update(cap_tensor=caps, cap_mask=cap_mask, final_loss=final.detach(), grad_names=np.array(gn),
                           grad_norms=torch.stack([params[n].grad.double().norm() for n in gn]))
            else:
                with torch.no_grad():
                    out, loss = model(dt, criterion, cc, "queries", eval_mode=True)
        rec.update(pred_logits=out["pred_logits"].detach(), pred_boxes=out["pred_boxes"].detach(), pred_count=out["pred_count"].detach())
        if "aux_outputs" in out and out["aux_outputs"]:
            rec.update(aux_pred_logits=out["aux_outputs"][0]["pred_logits"].detach(), aux_pred_boxes=out["aux_outputs"][0]["pred_boxes"].detach())
        if "seq" in out and isinstance(out["seq"], torch.Tensor):
            rec.update(seq=out["seq"], cap_prob_eval=out["caption_probs"]["cap_prob_eval"].detach())
            rec["has_seq"] = np.array(1)
        else:
            rec["has_seq"] = np.array(0)
        for i, (a, b) in enumerate(out["matched_indices"][0]):
            rec[f"match_{i}"] = torch.stack([a, b])
        for k, v in loss.items():
            rec[f"loss.{k}"] = torch.as_tensor(v).detach()
        save("pdvc_switch_" + name, **rec)


def make_init():
    """Seeded initialisation of the reference modules (MSDeformAttn._reset_parameters ms_deform_attn.py:62-77, the
    captioner variant :72, DeformableTransformer._reset_parameters deformable_transformer.py:54-63, PDVC.__init__
    pdvc.py:115-146): per-tensor (sum, sum of squares) after torch.manual_seed(0)."""
    rec = {}

    def stats(prefix, module):
        for k, v in module.state_dict().items():
            v = v.double()
            rec[f"{prefix}|{k}"] = np.array([float(v.sum()), float((v * v).sum()), float(v.numel())])

    torch.manual_seed(0)
    stats("msda", MSDeformAttn(64, 4, 8, 4))
    torch.manual_seed(0)
    stats("cap", MSDeformAttnCap(64, 4, 1, 4))
    from pdvc.deformable_transformer import DeformableTransformer
    torch.manual_seed(0)
    stats("transformer", DeformableTransformer(d_model=64, nhead=8, num_encoder_layers=2, num_decoder_layers=2,
                                                dim_feedforward=32, dropout=0.1, return_intermediate_dec=True,
                                                num_feature_levels=4, dec_n_points=4, enc_n_points=4))
    if FULL:
        opt, model, criterion, cc = build_pdvc()          # build_pdvc seeds with 0 before P.build
        stats("pdvc", model)
    save("init_stats", **rec)


if __name__ == "__main__":
    if "--only-pdvc" in sys.argv:
        make_pdvc()
        sys.exit(0)
    if "--only-init" in sys.argv:
        make_init()
        sys.exit(0)
    if "--only-collate" in sys.argv:
        make_collate()
        sys.exit(0)
    if "--only-anet-full" in sys.argv:
        make_anet_full()
        sys.exit(0)
    if "--only-yc2" in sys.argv:
        make_yc2()
        sys.exit(0)
    if "--only-gtprop" in sys.argv:
        make_gtprop()
        sys.exit(0)
    if "--only-train" in sys.argv:
        make_train()
        sys.exit(0)
    if "--only-posemb" in sys.argv:                   # the round-6 additions alone (the older fixtures are left untouched)
        make_module_cap3c()
        make_switches(only=("posemb",))
        sys.exit(0)
    for flag, fn in (("--only-dataset", make_dataset), ("--only-anet-c3d", make_anet_c3d), ("--only-f64", make_f64),
                     ("--only-anet-full-train", make_anet_full_train), ("--only-anet-full-b16", make_anet_full_b16),
                     ("--only-anet-full-train-b16", make_anet_full_train_b16), ("--only-yc2-train", make_yc2_train),
                     ("--only-full-train-probe", make_full_train_probe), ("--only-switches", make_switches)):
        if flag in sys.argv:
            fn()
            sys.exit(0)
    make_op()
    make_module()
    make_module_cap3c()
    make_matcher()
    make_collate()
    if FULL:
        make_pdvc()
        make_gtprop()
        make_yc2()
        make_anet_full()
        make_train()
        make_anet_c3d()
        make_f64()
        make_anet_full_train()
        make_anet_full_b16()
        make_anet_full_train_b16()
        make_yc2_train()
        make_full_train_probe()
        make_switches()
    make_dataset()
    make_init()
