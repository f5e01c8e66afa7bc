"""Model-level parity at the REAL dimensions of the BASELINE configs:
  * cfgs/anet_tsp_ssvg.yml training step (300 queries, vocabulary 8517): every loss term, matcher indices, the gradient
    norm of every parameter, selected gradients -- eager and through GraphedTrainStep (pdvc.py:540-620, train.py:403-406);
  * the eval forward's error measured against an fp64 evaluation of the reference, bounded by a small multiple of the
    REFERENCE's own fp32 error against that fp64 run (the committed noise floor, tests/golden/pdvc_anet_full_f64.npz);
  * cfgs/anet_c3d_ssvg.yml (feature_dim 500, BASELINE config 0) at its real dimensions on the GPU path."""
import numpy as np
import pytest
import torch

from helpers import load, path_census, pdvc_state, pdvc_dt, maxerr, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def to_dev(dt, dev=DEV):
    out = {}
    for k, v in dt.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.to(dev)
        elif k == "video_target":
            out[k] = [{a: b.to(dev) for a, b in t_.items()} for t_ in v]
        else:
            out[k] = v
    return out


def build_anet(train, **over):
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    f = load("pdvc_anet_full")
    opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda", **over)
    model, criterion, _, _ = build(opt)
    model.load_state_dict(pdvc_state(f, seed=100), strict=True)
    model = model.to(DEV)
    return f, opt, (model.train() if train else model.eval()), criterion


def train_batch(f, g):
    dt = to_dev(pdvc_dt(f, feat=int(f["feature_dim"]), seed=6))
    n_gt = [int(n) for n in g["n_gt"]]
    mx = max(n_gt)
    dt.update(cap_tensor=t(g["cap_tensor"]).to(DEV), cap_mask=t(g["cap_mask"]).to(DEV),
              gt_boxes_mask=torch.tensor([[k < n for k in range(mx)] for n in n_gt], dtype=torch.bool, device=DEV))
    return dt


def location_fed(name):
    """parameters whose gradient is a sum of sampling-LOCATION gradients (piecewise constant in the location)"""
    return any(k in name for k in ("sampling_offsets", "reference_points", "pos_trans", "query_embed"))


SMOOTH_TOL = 1e-3       # relative; the reference's own fp32-vs-fp64 deviation of every gradient norm is <= 7e-5 (fixture)
PERTURB = (1.0, 1.0 - 1e-6, 1.0 + 1e-6, 1.0 - 1e-5, 1.0 + 1e-5)


def train_grads(model, criterion, dt):
    model.zero_grad(set_to_none=True)
    out, loss = model(dt, criterion, None, "queries")
    wd = criterion.weight_dict
    final = sum(loss[k] * wd[k] for k in loss.keys() if k in wd)
    final.backward()
    grads = {n: p_.grad.detach().clone() for n, p_ in model.named_parameters() if p_.grad is not None}
    return out, {k: float(v) for k, v in loss.items()}, float(final), grads


def test_anet_full_dimension_train_step_matches_reference():
    """Every loss term, the matcher indices, the gradient norm of EVERY parameter and selected gradients element-wise.

    Smooth quantities must agree at the given input.  The sampling-location gradients are different in kind:
    d sample / d location is PIECEWISE CONSTANT in the location (difference of the two neighbouring frames, cuh:125-134),
    so when a sample sits on a frame boundary (this fixture has such samples) an upstream rounding difference of one
    fp32 ulp decides which side it falls on and moves the gradients that collect it (sampling_offsets, reference_points)
    by ~1 % -- measured: scaling the input features by (1 - 1e-6) moves transformer.reference_points.bias by 0.15 of
    16.3 and lands decoder.layers.1.cross_attn.sampling_offsets.bias exactly on the reference's value.  For the few
    parameters that miss the smooth tolerance at the given input the test therefore requires that the reference's value is
    reproduced under an input perturbation of at most 1e-5 relative (below the accuracy of the fp32 pipeline upstream),
    and that there are at most 8 such parameters, all fed by sampling locations."""
    g = load("pdvc_anet_full_train")
    f, opt, model, criterion = build_anet(True, transformer_dropout_prob=0.0, drop_prob=0.0)
    runs = []
    for scale in PERTURB:
        dt = train_batch(f, g)
        dt["video_tensor"] = dt["video_tensor"] * scale
        runs.append(train_grads(model, criterion, dt))
    out, loss, final, grads = runs[0]
    for k in [k for k in g if k.startswith("loss.")]:
        want = float(g[k])
        assert abs(loss[k[5:]] - want) <= 2e-4 * max(1.0, abs(want)), (k, loss[k[5:]], want)
    assert abs(final - float(g["final_loss"])) <= 2e-4 * float(g["final_loss"])
    for i, (a, b) in enumerate(out["matched_indices"][0]):
        assert torch.equal(torch.stack([a, b]), t(g[f"match_{i}"]))
    assert bool(g["match_same_in_f64"])
    names = [str(n) for n in g["grad_names"]]
    assert sorted(n for n, v in grads.items() if float(v.abs().max()) > 0) == \
        [n for n, w in zip(names, g["grad_norms"]) if float(w) > 0]
    sensitive, worst = [], 0.0
    for n, want in zip(names, g["grad_norms"]):
        want = float(want)
        errs = [abs(float(r[3][n].norm()) - want) / max(1e-3, want) for r in runs]
        if errs[0] > SMOOTH_TOL:
            # several boundary samples feed this parameter: its own value moves by `spread` under input perturbations
            # of <= 1e-5; the reference's value must lie within that band
            norms = [float(r[3][n].norm()) for r in runs]
            spread = (max(norms) - min(norms)) / max(1e-3, want)
            sensitive.append((n, errs[0], spread))
            # ONE boundary sample on the other side of a frame moves such a norm by up to ~1 % (measured: 0.15 of 16.3 for
            # reference_points.bias, 0.009 of 7.4 for this layer's sampling_offsets.bias); which side it takes is decided
            # by the summation order of the projection GEMM in front of it (library kernel vs gvl_proj_f32: both exact
            # fp32, different k order), which no input perturbation undoes -- hence the floor of 2e-2 for these
            assert location_fed(n) and min(errs) <= max(2e-2, 1.5 * spread), (n, errs, spread)
        else:
            worst = max(worst, errs[0])
    assert len(sensitive) <= 8 and all(location_fed(s_[0]) for s_ in sensitive), sensitive
    for k in [k for k in g if k.startswith("grad.") or k.startswith("grad_rows.")]:
        n, rows = (k[5:], slice(None)) if k.startswith("grad.") else (k[10:], None)
        step = {"caption_head.0.logit.weight": 97, "query_embed.weight": 13}.get(n)
        scale = max(1e-3, float(np.abs(g[k]).max()))
        errs = [maxerr(r[3][n][::step] if step else r[3][n], g[k]) / scale for r in runs]
        if any(n == s_[0] for s_ in sensitive) or location_fed(n):
            continue                                   # covered by the norm band above
        # (gradients next to a boundary sample inherit a little of its step -- 1.2e-3 here, the same figure with the
        #  library projection GEMM, where a 1e-6 input perturbation happens to undo it, and with gvl_proj_f32, where it
        #  does not: element-wise within 5e-3 at the given input)
        assert errs[0] <= 5e-3, (k, errs)
    print(f"worst relative gradient-norm error of the smooth parameters {worst:.2e}; boundary-sensitive: {sensitive}")


def test_anet_full_dimension_train_step_at_the_headline_batch_matches_reference():
    """The same comparison at B = 16 (the benchmark's batch: 16 videos of different valid lengths, 0..10 events each, 61 caption
    rows; tests/golden/pdvc_anet_full_train_b16.npz, generated by importing the reference): every loss term, the matcher indices
    of every video, the gradient norm of EVERY parameter.  Boundary-sensitive (sampling-location-fed) parameters as above; with
    eight times the samples more of them see a boundary sample, so their band is checked, not their count."""
    g = load("pdvc_anet_full_train_b16")
    f, opt, model, criterion = build_anet(True, transformer_dropout_prob=0.0, drop_prob=0.0)
    runs = []
    for scale in PERTURB:
        dt = to_dev(pdvc_dt(g, feat=int(g["feature_dim"]), seed=16))
        n_gt = [int(n) for n in g["n_gt"]]
        mx = max(n_gt)
        dt.update(cap_tensor=t(g["cap_tensor"]).to(DEV), cap_mask=t(g["cap_mask"]).to(DEV),
                  gt_boxes_mask=torch.tensor([[k < n for k in range(mx)] for n in n_gt], dtype=torch.bool, device=DEV))
        dt["video_tensor"] = dt["video_tensor"] * scale
        if scale == 1.0:
            res, ran = path_census(lambda: train_grads(model, criterion, dt))
            runs.append(res)
            # what served the step: every Linear of the 2 + 2 layers forward and input gradient on
            # gvl_linear_f16x3_f32, its weight + bias gradient on gvl_wgrad_f16x3_f32, both decoder self-attentions on the
            # attention-core kernels (forward, dk / dv, dq each), four deformable-attention launches each way
            assert ran["layer_gemm"] >= 2 * 24 and ran["wgrad_f16x3"] >= 24 and ran["mha_train"] == 6, ran
            assert ran["fwd_t1d_d64"] == 4 and ran["bwd_t1d_d64"] == 4 and ran["layer_norm_etc"] >= 3 * 10, ran
        else:
            runs.append(train_grads(model, criterion, dt))
    out, loss, final, grads = runs[0]
    for k in [k for k in g if k.startswith("loss.")]:
        want = float(g[k])
        if np.isnan(want):          # (loss_self_iou: the reference's mean over the pairs of a video WITHOUT events is NaN; unweighted)
            assert np.isnan(loss[k[5:]]) and k[5:] not in criterion.weight_dict, k
            continue
        assert abs(loss[k[5:]] - want) <= 2e-4 * max(1.0, abs(want)), (k, loss[k[5:]], want)
    assert abs(final - float(g["final_loss"])) <= 2e-4 * float(g["final_loss"])
    assert len(out["matched_indices"][0]) == 16
    for i, (a, b) in enumerate(out["matched_indices"][0]):
        assert torch.equal(torch.stack([a, b]), t(g[f"match_{i}"])), i
    names = [str(n) for n in g["grad_names"]]
    assert sorted(n for n, v in grads.items() if float(v.abs().max()) > 0) == \
        [n for n, w in zip(names, g["grad_norms"]) if float(w) > 0]
    sensitive, worst = [], 0.0
    for n, want in zip(names, g["grad_norms"]):            # (this fixture's norms are accumulated in float64; so are ours)
        want = float(want)
        errs = [abs(float(r[3][n].double().norm()) - want) / max(1e-3, want) for r in runs]
        if errs[0] > SMOOTH_TOL:
            norms = [float(r[3][n].double().norm()) for r in runs]
            spread = (max(norms) - min(norms)) / max(1e-3, want)
            sensitive.append((n, errs[0], spread))
            assert location_fed(n) and min(errs) <= max(2e-2, 1.5 * spread), (n, errs, spread)
        else:
            worst = max(worst, errs[0])
    assert all(location_fed(s_[0]) for s_ in sensitive), sensitive
    for k, step in (("grad_rows.caption_head.0.logit.weight", 97), ("grad.class_head.1.weight", None),
                    ("grad.count_head.0.bias", None), ("grad.transformer.level_embed", None)):
        n = k.split(".", 1)[1]
        got = grads[n][::step] if step else grads[n]
        assert maxerr(got, g[k]) <= 5e-3 * max(1e-3, float(np.abs(g[k]).max())), k
    print(f"B = 16: worst relative gradient-norm error of the smooth parameters {worst:.2e}; boundary-sensitive: {sensitive}")


def test_yc2_long_video_train_step_matches_reference():
    """BASELINE config 4's model (cfgs/yc2_tsn_dvc.yml: 3072-d input, 100 queries, T = 512 -> S = 960) through ONE training
    forward / backward on B = 8 videos against the reference (tests/golden/pdvc_yc2_train.npz): losses, matcher indices, the float64-accumulated
    gradient norm of every parameter.  The deformable-attention backward of this step is the long-video form of round 4
    (k_bwd_t1d_own: level 0 in global memory, rows owned across query chunks) -- asserted -- so this pins it inside the real
    model, not only at op level.  Location-fed parameters as in the tests above (at T = 512 d sample / d location = T_l dv is
    five times steeper than at T = 100: their band is what is checked)."""
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    g = load("pdvc_yc2_train")
    opt = make_opt("yc2_tsn_dvc", max_caption_len=8, frame_embedding_num=512, device="cuda", transformer_dropout_prob=0.0,
                   drop_prob=0.0)
    model, criterion, _, _ = build(opt)
    model.load_state_dict(pdvc_state(g, seed=512), strict=True)
    model = model.to(DEV).train()
    runs = []
    for scale in PERTURB:
        dt = to_dev(pdvc_dt(g, feat=int(g["feature_dim"]), seed=4))
        n_gt = [int(n) for n in g["n_gt"]]
        mx = max(n_gt)
        dt.update(cap_tensor=t(g["cap_tensor"]).to(DEV), cap_mask=t(g["cap_mask"]).to(DEV),
                  gt_boxes_mask=torch.tensor([[k < n for k in range(mx)] for n in n_gt], dtype=torch.bool, device=DEV))
        dt["video_tensor"] = dt["video_tensor"] * scale
        runs.append(train_grads(model, criterion, dt))
    # the backward form this batch is served by (autograd runs the op on its own thread; the same shapes on this one)
    from gvl_amd import _lib
    from gvl_amd.deformable_transformer import make_level_tensors
    from gvl_amd.ops.modules.ms_deform_attn import temporal_shapes_2d
    tsh, lsi = make_level_tensors([512, 256, 128, 64], torch.device(DEV))
    Bv = len(n_gt)
    MSDA.msda1d_fused_backward(torch.randn(Bv, 960, 8, 64, device=DEV), temporal_shapes_2d(tsh, lsi), lsi,
                               torch.randn(Bv, 960, 256, device=DEV), torch.rand(Bv, 960, 4, 1, device=DEV),
                               torch.randn(Bv, 960, 512, device=DEV), 4, 4)
    assert _lib.lib().gvl_msda_last_kernel().decode() == "k_bwd_t1d_own"
    out, loss, final, grads = runs[0]
    for k in [k for k in g if k.startswith("loss.")]:
        want = float(g[k])
        if np.isnan(want):          # (loss_self_iou of a batch that has a video without events: NaN in the reference too; unweighted)
            assert np.isnan(loss[k[5:]]) and k[5:] not in criterion.weight_dict, k
            continue
        assert abs(loss[k[5:]] - want) <= 5e-4 * max(1.0, abs(want)), (k, loss[k[5:]], want)
    assert abs(final - float(g["final_loss"])) <= 5e-4 * float(g["final_loss"])
    for i, (a, b) in enumerate(out["matched_indices"][0]):
        assert torch.equal(torch.stack([a, b]), t(g[f"match_{i}"])), i
    names = [str(n) for n in g["grad_names"]]
    assert sorted(n for n, v in grads.items() if float(v.abs().max()) > 0) == \
        [n for n, w in zip(names, g["grad_norms"]) if float(w) > 0]
    # The yardstick at T = 512 is the reference's OWN fp32 error, recorded in the fixture from its float64 evaluation of the same
    # step: sampling-location gradients are differences of neighbouring frames times T_l, and every parameter upstream of an
    # offsets projection inherits their noise -- the reference's fp32 gradients sit 0.1 ... 30 % (captioner's sampling offsets)
    # away from its fp64 ones here.  A parameter passes if its norm is within max(2e-3, 3 x that error) of the reference's fp32
    # value at the given input, or reproduces it inside its own band under input perturbations of <= 1e-5.
    assert bool(g["match_same_in_f64"])
    ref_err = {n: float(e) / max(1e-3, float(w)) for n, e, w in zip(names, g["grad_norm_f32_err"], g["grad_norms_f64"])}
    sensitive, worst, worst_ratio = [], 0.0, 0.0
    for n, want in zip(names, g["grad_norms"]):
        want = float(want)
        tol = max(2 * SMOOTH_TOL, 3.0 * ref_err[n])
        errs = [abs(float(r[3][n].double().norm()) - want) / max(1e-3, want) for r in runs]
        worst_ratio = max(worst_ratio, errs[0] / max(ref_err[n], 1e-6))
        if errs[0] > tol:
            norms = [float(r[3][n].double().norm()) for r in runs]
            spread = (max(norms) - min(norms)) / max(1e-3, want)
            sensitive.append((n, errs[0], spread, ref_err[n]))
            assert min(errs) <= max(tol, 1.5 * spread), (n, errs, spread, ref_err[n])
        else:
            worst = max(worst, errs[0])
    assert len(sensitive) <= 6, sensitive
    for k in [k for k in g if k.startswith("grad.")]:
        n = k[5:]
        got = grads[n] if grads[n].numel() <= 4096 else grads[n][::16, ::8]
        assert maxerr(got, g[k]) <= max(5e-3, 3.0 * ref_err[n]) * max(1e-3, float(np.abs(g[k]).max())), k
    print(f"yc2 T = 512: worst relative gradient-norm error inside tolerance {worst:.2e}; worst (our error) / (reference's own fp32 "
          f"error) {worst_ratio:.2f}; outside at the given input: {sensitive}")


def test_anet_full_dimension_graphed_train_step_equals_eager():
    """the captured, layout-independent step (padded targets with capacities larger than the batch) computes what the
    eager step computes at the real dimensions: losses and smooth gradients agree to rounding.  The captioner's GEMMs run
    on a different number of rows (compact padded row set vs exact), i.e. through different library kernels, and with
    ~6e5 samples per layer some always sit within an ulp-sized step of a frame boundary -- gradients that collect
    sampling-location gradients (see test_anet_full_dimension_train_step_matches_reference), and to a lesser degree
    everything upstream of them, therefore only agree to ~1 % here.  The sharp (1e-4) graphed == eager comparison over
    many batch layouts is tests/test_gpu_layout_independent.py, at dimensions where boundary samples are rare."""
    from gvl_amd.parallel import GraphedTrainStep
    g = load("pdvc_anet_full_train")
    kw = dict(transformer_dropout_prob=0.0, drop_prob=0.0, lr=1e-10, weight_decay=0.0, grad_clip=1e9)
    f, opt, model_a, crit_a = build_anet(True, **kw)
    _, _, model_b, crit_b = build_anet(True, **kw)
    dt = train_batch(f, g)
    _, loss_a, final_a, grads_a = train_grads(model_a, crit_a, dt)
    step = GraphedTrainStep(model_b, crit_b, opt, max_gt=10, max_cap_len=20, max_events=40)
    for _ in range(2):
        final_b, loss_b = step(dt)
        assert step.captures == 1 and len(step.graphs) == 1
        assert abs(float(final_b) - final_a) <= 1e-5 * abs(final_a)
        for k in loss_a:
            assert abs(float(loss_b[k]) - loss_a[k]) <= 1e-5 * max(1.0, abs(loss_a[k])), k
        for n, p_ in model_b.named_parameters():
            if n in grads_a:
                na, nb = float(grads_a[n].norm()), float(p_.grad.norm())
                assert abs(na - nb) <= (3e-2 if location_fed(n) else 5e-3) * max(1e-3, na), (n, na, nb)


def test_anet_full_eval_error_is_at_the_fp32_noise_floor():
    """|gvl_amd fp32 - reference fp64| <= 4 x |reference fp32 - reference fp64| (+ 2e-6): gvl_amd's model-level outputs
    are as close to the exact values as the reference's own fp32 run is.  Greedy tokens: the reference's fp32 and fp64
    runs agree on every token; gvl_amd may flip a near-tie (different summation order; measured: 1 of 18 000 tokens):
    at least 99.95 % of the tokens agree."""
    h = load("pdvc_anet_full_f64")
    f, opt, model, criterion = build_anet(False)
    dt = to_dev(pdvc_dt(f, feat=int(f["feature_dim"]), seed=6))
    with torch.no_grad():
        out, _ = model(dt, criterion, None, "queries", eval_mode=True)
    report = {}
    for k in ("pred_logits", "pred_boxes", "pred_count"):
        err, floor = maxerr(out[k].double(), h[k + "_f64"]), float(h[k + "_f32_err"])
        report[k] = (err, floor)
        assert err <= 4.0 * floor + 2e-6, (k, err, floor)
    assert float(h["seq_agree"]) == 1.0
    seq = out["seq"].cpu()
    agree = float((seq == t(h["seq_f64"])).float().mean())
    report["tokens"] = agree
    assert agree >= 0.9995
    print("error vs fp64 (gvl_amd, reference fp32 floor):", report)


def test_anet_tsp_msvg_dvc_config_runs_the_headline_model():
    """BASELINE config 4 names cfgs/anet_tsp_msvg_dvc.yml (the 8-GPU data-parallel case).  Against cfgs/anet_tsp_ssvg.yml that file
    changes three switches of the frozen-RoBERTa text branch only (enable_layer_diff_text_feature, enable_sentence_context_modeling,
    enable_sentence_pos_embedding: SURVEY section 2 #15, out of scope) -- on the hot path it IS the headline model: the same
    parameters, and the reference-generated B = 16-class eval golden holds under its name, eager and through the captured forward
    that the data-parallel eval shards use."""
    from gvl_amd.config import make_opt
    from gvl_amd.parallel import GraphedEvalForward, shard_batch
    from gvl_amd.pdvc import build
    f = load("pdvc_anet_full")
    opt = make_opt("anet_tsp_msvg_dvc", num_queries=300, frame_embedding_num=100, device="cuda")
    ref_opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
    assert {k: v for k, v in vars(opt).items() if k != "id"} == {k: v for k, v in vars(ref_opt).items() if k != "id"}
    model, criterion, _, _ = build(opt)
    model.load_state_dict(pdvc_state(f, seed=100), strict=True)
    model = model.to(DEV).eval()
    dt = to_dev(pdvc_dt(f, feat=int(f["feature_dim"]), seed=6))
    graphed = GraphedEvalForward(model, criterion)
    with torch.no_grad():
        out, _ = model(dt, criterion, None, "queries", eval_mode=True)
        out_g, _ = graphed(dt)
    for o in (out, out_g):
        assert maxerr(o["pred_boxes"], f["pred_boxes"]) <= 2e-4
        assert maxerr(o["pred_logits"], f["pred_logits"]) <= 3e-4
        assert maxerr(o["pred_count"], f["pred_count"]) <= 3e-4
    # the shard a rank of a 2-way data-parallel eval takes reproduces its videos' rows of the full batch
    B = dt["video_tensor"].shape[0]
    if B >= 2:
        sh = shard_batch(dt, 1, 2)
        with torch.no_grad():
            out_s, _ = model(sh, criterion, None, "queries", eval_mode=True)
        assert maxerr(out_s["pred_boxes"], out["pred_boxes"][1::2]) <= 2e-5
        assert maxerr(out_s["pred_logits"], out["pred_logits"][1::2]) <= 2e-5


def test_anet_c3d_config_eval_matches_reference():
    """BASELINE config 0 (cfgs/anet_c3d_ssvg.yml: 500-d C3D features, 30 queries) on the GPU path.  The reference runs
    this config through its CPU fallback; gvl_amd has no CPU path by design, the model is the same."""
    from gvl_amd.config import make_opt
    from gvl_amd.parallel import GraphedEvalForward
    from gvl_amd.pdvc import build
    f = load("pdvc_anet_c3d")
    opt = make_opt("anet_c3d_ssvg", device="cuda")
    assert (opt.feature_dim, opt.num_queries, opt.vocab_size) == (int(f["feature_dim"]), int(f["num_queries"]), int(f["vocab_size"])) == (500, 30, 8517)
    model, criterion, _, _ = build(opt)
    model.load_state_dict(pdvc_state(f, seed=500), strict=True)
    model = model.to(DEV).eval()
    dt = to_dev(pdvc_dt(f, feat=500, seed=8))
    graphed = GraphedEvalForward(model, criterion)
    for mode in ("eager", "graph"):
        with torch.no_grad():
            out, loss = model(dt, criterion, None, "queries", eval_mode=True) if mode == "eager" else graphed(dt)
        assert maxerr(out["pred_boxes"], f["pred_boxes"]) <= 1e-4
        assert maxerr(out["aux_outputs"][0]["pred_boxes"], f["aux_pred_boxes"]) <= 1e-4
        assert maxerr(out["pred_logits"], f["pred_logits"]) <= 3e-4
        assert maxerr(out["pred_count"], f["pred_count"]) <= 3e-4
        for i in range(len(out["matched_indices"][0])):
            assert torch.equal(torch.stack(out["matched_indices"][0][i]), t(f[f"match_{i}"]))
        for k in ("loss_ce", "loss_giou", "loss_counter", "loss_self_iou"):
            assert maxerr(loss[k].reshape(()), f[f"loss.{k}"].reshape(())) <= 3e-4 * max(1.0, abs(float(f[f"loss.{k}"]))), k
        seq, ref_seq = out["seq"].cpu(), t(f["seq"])
        assert seq.shape == ref_seq.shape
        assert float((seq == ref_seq).float().mean()) >= 0.95


def test_replayed_step_rebuilds_every_operand_plane():
    """A captured train step must re-split EVERY weight operand from the parameters' current values on every replay (the planes of
    the model's TrainPlanes, including the mirrored column block of the LSTM's W_ih): a forward that records no refresh would
    multiply by the planes of the capture-time weights for ever -- silently, a few per cent off (seen once, round 6: the
    data-parallel replicas drifted from the serial step).  Checked where no training noise can be mistaken for it: the planes
    are overwritten with garbage, the step is replayed at a learning rate of 0, and afterwards every plane buffer must hold
    exactly what an eager refresh() of the same parameters writes."""
    from gvl_amd.parallel import GraphedTrainStep
    g = load("pdvc_anet_full_train")
    kw = dict(transformer_dropout_prob=0.0, drop_prob=0.0, lr=0.0, weight_decay=0.0, grad_clip=1e9)
    f, opt, model, crit = build_anet(True, **kw)
    dt = train_batch(f, g)
    step = GraphedTrainStep(model, crit, opt, max_gt=10, max_cap_len=20, max_events=40)
    step(dt)
    step(dt)
    assert step.captures == 1 and step.replays >= 1
    tp = model.train_planes()
    assert len(tp.operands) > 20 and len(tp.mirrors) >= 1
    tp.refresh()
    torch.cuda.synchronize()
    bufs = [t_ for op_ in tp.operands for pl in (op_[2], op_[3]) for t_ in (pl.hi, pl.lo, pl.scale)]
    want = [t_.clone() for t_ in bufs]
    for t_ in bufs:
        t_.fill_(7.0) if t_.dtype == torch.float32 else t_.view(torch.int16).fill_(0x3C01)
    for buf, _ in tp.mirrors:
        buf.fill_(123.0)
    step(dt)                                                          # (lr = 0: the parameters are what refresh() above saw)
    torch.cuda.synchronize()
    assert step.captures == 1
    for i, (a, b) in enumerate(zip(bufs, want)):
        assert torch.equal(a, b), i
    for buf, src in tp.mirrors:
        assert torch.equal(buf, src().detach())
