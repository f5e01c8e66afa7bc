"""CPU: pin the oracle (oracle/msda_ref.c + oracle/torch_ref.py) against the golden vectors that
tests/golden/make_golden.py produced by importing the reference.  No GPU, no reference access."""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load, t, module_state, pdvc_state, pdvc_dt, maxerr
from oracle import msda_oracle as O
from oracle import torch_ref as R

OP_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "op_*.npz")))


@pytest.mark.parametrize("case", OP_CASES)
@pytest.mark.parametrize("pad", ["zeros", "border"])
def test_c_oracle_op_matches_reference(case, pad):
    f = load(case)
    tol = 1e-11 if f["value"].dtype == np.float64 else 2e-5
    scale = max(1.0, float(np.abs(f[f"gloc_{pad}"]).max()))
    out = O.msda_forward(f["value"], f["shapes"], f["lsi"], f["loc"], f["aw"], pad)
    assert maxerr(out, f[f"out_{pad}"]) <= tol
    gv, gl, gw = O.msda_backward(f["value"], f["shapes"], f["lsi"], f["loc"], f["aw"], f["gout"], pad)
    assert maxerr(gv, f[f"gvalue_{pad}"]) <= tol
    assert maxerr(gl, f[f"gloc_{pad}"]) <= tol * scale
    assert maxerr(gw, f[f"gaw_{pad}"]) <= tol * scale


@pytest.mark.parametrize("case", OP_CASES)
def test_c_oracle_sample_matches_reference(case):
    f = load(case)
    tol = 1e-11 if f["value"].dtype == np.float64 else 2e-5
    s = O.msda_sample(f["value"], f["shapes"], f["lsi"], f["loc"], "border")
    assert maxerr(s, f["sample_border"]) <= tol


@pytest.mark.parametrize("case", OP_CASES)
@pytest.mark.parametrize("pad", ["zeros", "border"])
def test_torch_oracle_core_matches_reference(case, pad):
    f = load(case)
    tol = 1e-11 if f["value"].dtype == np.float64 else 2e-5
    out = R.msda_core(t(f["value"]), t(f["shapes"]), t(f["loc"]), t(f["aw"]), pad)
    assert maxerr(out, f[f"out_{pad}"]) <= tol


def test_grad_loc_y_identity_h1():
    """H=1, y=0.5: zeros-mode grad_loc_y = -w * grad_w; border-mode grad_loc_y = 0."""
    f = load("op_t1d_d8_f64")
    assert maxerr(f["gloc_zeros"][..., 1], -f["aw"] * f["gaw_zeros"]) < 1e-11
    assert float(np.abs(f["gloc_border"][..., 1]).max()) == 0.0


@pytest.mark.parametrize("refdim", [1, 2])
@pytest.mark.parametrize("pad", ["zeros", "border"])
def test_torch_oracle_module(refdim, pad):
    from helpers import synth_array
    f = load(f"module_ref{refdim}")
    B, Q, C, M, L, P, _ = [int(v) for v in f["meta"]]
    S = int(f["tshapes"].sum())
    sd = module_state("attn.", seed=100 + refdim)
    query = t(synth_array(f"mod{refdim}.query", (B, Q, C), 1)).requires_grad_()
    inp = t(synth_array(f"mod{refdim}.input", (B, S, C), 1)).requires_grad_()
    ref = t(synth_array(f"mod{refdim}.ref", (B, Q, L, refdim), 1, 0.05, 0.95))
    if refdim == 2:
        ref[..., 1] = ref[..., 1] * 0.5
    gout = t(synth_array(f"mod{refdim}.gout", (B, Q, C), 1))
    out = R.msda_module(sd, "attn.", query, ref, inp, t(f["tshapes"]), t(f["mask"]), M, L, P, pad_mode=pad)
    assert maxerr(out, f[f"out_{pad}"]) < 5e-5
    out.backward(gout)
    assert maxerr(query.grad, f[f"gquery_{pad}"]) < 5e-4
    assert maxerr(inp.grad, f[f"ginput_{pad}"]) < 5e-4


def test_torch_oracle_cap_module():
    from helpers import synth_array
    f = load("module_cap")
    B, Q, C, M, L, P, _ = [int(v) for v in f["meta"]]
    S = int(f["tshapes"].sum())
    sd = module_state("cap.", M=1, qdim=2 * C, seed=200)
    query = t(synth_array("cap.query", (B, Q, 2 * C), 1))
    inp = t(synth_array("cap.input", (B, S, C), 1))
    ref = t(synth_array("cap.ref", (B, Q, L, 2), 1, 0.05, 0.95))
    ref[..., 1] *= 0.5
    out = R.msda_module(sd, "cap.", query, ref, inp, t(f["tshapes"]), t(f["mask"]), 1, L, P, cap=True)
    assert maxerr(out, f["out"]) < 5e-5


@pytest.mark.parametrize("tag,pad", [("cuda", "zeros"), ("cpu", "border")])
def test_torch_oracle_pdvc_eval(tag, pad):
    f = load("pdvc_eval")
    sd = pdvc_state(f)
    dt = pdvc_dt(f)
    with torch.no_grad():
        out = R.pdvc_eval_forward(sd, dt, pad_mode=pad, max_caption_len=6)
    assert maxerr(out["memory"], f[f"{tag}.memory"]) < 2e-4
    assert maxerr(out["hs"], f[f"{tag}.hs"]) < 5e-4
    assert maxerr(out["inter_references"], f[f"{tag}.inter_references"]) < 1e-4
    assert maxerr(out["pred_logits"], f[f"{tag}.pred_logits"]) < 5e-4
    assert maxerr(out["pred_boxes"], f[f"{tag}.pred_boxes"]) < 1e-4
    assert maxerr(out["pred_count"], f[f"{tag}.pred_count"]) < 5e-4
    assert maxerr(out["aux_boxes"][0], f[f"{tag}.aux_pred_boxes"]) < 1e-4
    assert torch.equal(out["seq"], t(f[f"{tag}.seq"]))
    assert maxerr(out["cap_prob_eval"], f[f"{tag}.cap_prob_eval"]) < 5e-4
    # matcher index path: bit-exact
    tg = dt["video_target"]
    C = R.matcher_cost(out["pred_logits"], out["pred_boxes"], torch.cat([x["labels"] for x in tg]),
                       torch.cat([x["boxes"] for x in tg]))
    idx, rl = R.hungarian(C, [len(x["boxes"]) for x in tg])
    for i in range(len(tg)):
        assert torch.equal(torch.stack(idx[i]), t(f[f"{tag}.match_{i}"]))
        assert torch.equal(torch.stack(rl[i]), t(f[f"{tag}.rl_match_{i}"]))


@pytest.mark.parametrize("name,seed,dt_seed,cap_len", [("pdvc_anet_full", 100, 6, 30), ("pdvc_yc2", 512, 4, 8)])
def test_torch_oracle_full_dimension_configs(name, seed, dt_seed, cap_len):
    """the CPU restatement at the REAL dimensions of BASELINE configs 1-2 (anet: 300 queries, vocabulary 8517) and
    4 (yc2: 3072-d input, T = 512) against the reference's own run of those configs"""
    f = load(name)
    sd = pdvc_state(f, seed=seed)
    dt = pdvc_dt(f, feat=int(f["feature_dim"]), seed=dt_seed)
    with torch.no_grad():
        out = R.pdvc_eval_forward(sd, dt, pad_mode="zeros", max_caption_len=cap_len)
    stride = 4 if name == "pdvc_anet_full" else 8
    assert maxerr(out["memory"][:, ::stride], f["memory_rows"]) < 2e-4 * max(1.0, float(np.abs(f["memory_rows"]).max()))
    assert maxerr(out["pred_boxes"], f["pred_boxes"]) < 5e-4
    assert maxerr(out["pred_logits"], f["pred_logits"]) < 5e-3
    assert maxerr(out["pred_count"], f["pred_count"]) < 5e-3
    seq, ref_seq = out["seq"].reshape(-1, out["seq"].shape[-1]), t(f["seq"]).reshape(-1, f["seq"].shape[-1])
    n = min(seq.shape[1], ref_seq.shape[1])
    assert float((seq[:, :n] == ref_seq[:, :n]).float().mean()) >= 0.95
    tg = dt["video_target"]
    C = R.matcher_cost(out["pred_logits"], out["pred_boxes"], torch.cat([x["labels"] for x in tg]),
                       torch.cat([x["boxes"] for x in tg]))
    idx, _ = R.hungarian(C, [len(x["boxes"]) for x in tg])
    for i in range(len(tg)):
        assert torch.equal(torch.stack(idx[i]), t(f[f"match_{i}"]))


def test_torch_oracle_headline_batch_of_16():
    """the CPU restatement on the HEADLINE batch (B = 16 videos of cfgs/anet_tsp_ssvg.yml, different valid lengths, 0..10
    events per video) against the reference's run (tests/golden/pdvc_anet_full_b16.npz).  The captioner is cut to its first
    8 token steps here to keep the CPU suite short (all 30 are compared on the GPU path, tests/test_gpu_model.py)."""
    f = load("pdvc_anet_full_b16")
    sd = pdvc_state(f, seed=100)
    dt = pdvc_dt(f, feat=int(f["feature_dim"]), seed=16)
    with torch.no_grad():
        out = R.pdvc_eval_forward(sd, dt, pad_mode="zeros", max_caption_len=8)
    assert maxerr(out["pred_boxes"], f["pred_boxes"]) < 5e-4
    assert maxerr(out["pred_logits"], f["pred_logits"]) < 5e-3
    assert maxerr(out["pred_count"], f["pred_count"]) < 5e-3
    seq, ref_seq = out["seq"].reshape(-1, out["seq"].shape[-1]), t(f["seq"].astype(np.int64)).reshape(-1, f["seq"].shape[-1])
    n = min(seq.shape[1], ref_seq.shape[1])
    assert n >= 8 and float((seq[:, :n] == ref_seq[:, :n]).float().mean()) >= 0.95
    tg = dt["video_target"]
    C = R.matcher_cost(out["pred_logits"], out["pred_boxes"], torch.cat([x["labels"] for x in tg]),
                       torch.cat([x["boxes"] for x in tg]))
    idx, _ = R.hungarian(C, [len(x["boxes"]) for x in tg])
    for i in range(len(tg)):
        assert torch.equal(torch.stack(idx[i]), t(f[f"match_{i}"]))


def test_torch_oracle_captioner_step():
    f = load("captioner_step")
    sd = pdvc_state(load("pdvc_eval"))
    with torch.no_grad():
        logp, (h1, c1) = R.captioner_step(sd, "caption_head.1.", t(f["it"]), (t(f["h0"]), t(f["c0"])), t(f["hs"]),
                                          t(f["ref_in"]), t(f["memory"]), t(f["tshapes"]), t(f["mask"]))
    assert maxerr(logp, f["logp"]) < 2e-4
    assert maxerr(h1, f["h1"]) < 5e-5
    assert maxerr(c1, f["c1"]) < 5e-5


def test_torch_oracle_matcher_model_fixture():
    f = load("matcher_model")
    sizes = [int(s) for s in f["sizes"]]
    C = R.matcher_cost(t(f["logits"]), t(f["boxes"]), torch.zeros(sum(sizes), dtype=torch.long), t(f["tgt_boxes"]),
                       w_cl=2.0, cl_match_mats=0)
    for i, c in enumerate(C.split(sizes, -1)):
        assert torch.equal(c[i], t(f[f"C_{i}"]))          # same fp32 op sequence -> bit-identical cost
    idx, rl = R.hungarian(C, sizes)
    for i in range(len(sizes)):
        assert torch.equal(torch.stack(idx[i]), t(f[f"idx_{i}"]))
        assert torch.equal(torch.stack(rl[i]), t(f[f"rl_{i}"]))


def test_scipy_lsap_contract():
    """the golden (cost -> indices) pairs were produced with scipy 1.15.3; the oracle solver must reproduce them"""
    from scipy.optimize import linear_sum_assignment
    f = load("lsap_cases")
    names = sorted({k.split(".")[0] for k in f if "." in k})
    assert len(names) >= 10
    for n in names:
        r, c = linear_sum_assignment(f[f"{n}.C"])
        assert np.array_equal(r, f[f"{n}.rows"]) and np.array_equal(c, f[f"{n}.cols"]), n


def test_seeded_initialisation_matches_reference():
    """gvl_amd modules consume the RNG in the reference's order and apply the same special initialisations
    (offset-bias grid, zeroed attention weights, xavier passes, box-head biases): per-tensor (sum, sum of squares)
    equal the reference's after torch.manual_seed(0).  Construction needs no GPU."""
    import warnings
    from gvl_amd.ops.modules import MSDeformAttn, MSDeformAttnCap
    from gvl_amd.deformable_transformer import DeformableTransformer
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    f = load("init_stats")

    def check(prefix, module):
        sd = module.state_dict()
        keys = [k.split("|", 1)[1] for k in f if k.startswith(prefix + "|")]
        assert sorted(keys) == sorted(sd.keys())
        for k in keys:
            v = sd[k].double()
            want = f[f"{prefix}|{k}"]
            assert v.numel() == int(want[2]), k
            assert abs(float(v.sum()) - want[0]) <= 1e-6 * max(1.0, abs(want[0])), (prefix, k)
            assert abs(float((v * v).sum()) - want[1]) <= 1e-6 * max(1.0, abs(want[1])), (prefix, k)

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(0)
        check("msda", MSDeformAttn(64, 4, 8, 4))
        torch.manual_seed(0)
        check("cap", MSDeformAttnCap(64, 4, 1, 4))
        torch.manual_seed(0)
        check("transformer", DeformableTransformer(d_model=64, nhead=8, num_encoder_layers=2, num_decoder_layers=2,
                                                    dim_feedforward=32, dropout=0.1, return_intermediate_dec=True,
                                                    num_feature_levels=4, dec_n_points=4, enc_n_points=4))
        opt = make_opt(num_queries=8, feature_dim=64, vocab_size=40, max_caption_len=6, device="cpu")
        torch.manual_seed(0)
        model, _, _, _ = build(opt)
        check("pdvc", model)


def test_postprocess_matches_reference():
    """gvl_amd.postprocess.PostProcess (host glue, runs anywhere) on the reference's own eval outputs reproduces the
    reference's PostProcess.forward (pdvc.py:1003-1089): ranking, clipped + scaled boxes, caption order and scores."""
    from gvl_amd.config import make_opt
    from gvl_amd.postprocess import PostProcess
    f = load("pdvc_eval")
    out = {"pred_logits": t(f["cuda.pred_logits"]), "pred_boxes": t(f["cuda.pred_boxes"]),
           "pred_count": t(f["cuda.pred_count"]), "seq": t(f["cuda.seq"]),
           "caption_probs": {"cap_prob_eval": t(f["cuda.cap_prob_eval"])}}

    class _Tr:
        @staticmethod
        def rtranslate(s):
            return " ".join(str(int(x)) for x in s if x > 0)
    loader = type("L", (), {"dataset": type("D", (), {"translator": _Tr})})
    dt = pdvc_dt(f)
    res = PostProcess(make_opt(device="cpu"))(out, dt["video_length"][:, 1], loader)
    assert len(res) == 2
    for i, r in enumerate(res):
        assert maxerr(r["scores"], f[f"post.{i}.scores"]) < 1e-6
        assert torch.equal(r["labels"], t(f[f"post.{i}.labels"]))
        assert torch.equal(r["query_id"], t(f[f"post.{i}.query_id"]))
        assert torch.equal(r["raw_idx"], t(f[f"post.{i}.raw_idx"]))
        assert maxerr(r["boxes"], f[f"post.{i}.boxes"]) < 1e-4
        assert int(r["pred_seq_len"]) == int(f[f"post.{i}.pred_seq_len"])
        assert list(r["captions"]) == [str(c) for c in f[f"post.{i}.captions"]]
        assert np.allclose(np.asarray(r["caption_scores"], np.float64), f[f"post.{i}.caption_scores"], atol=1e-5)


def test_result_json_wire_format(tmp_path):
    """gvl_amd.eval_utils: PostProcess results -> the reference's result-file records (eval_utils.py:216-239,136-141).
    Built from the PostProcess golden (reference outputs), so field names, filtering and value types are the
    reference's."""
    import json
    from gvl_amd import eval_utils as E
    f = load("pdvc_eval")
    results = []
    for i in range(2):
        results.append({"scores": t(f[f"post.{i}.scores"]), "labels": t(f[f"post.{i}.labels"]),
                        "boxes": t(f[f"post.{i}.boxes"]), "raw_boxes": t(f[f"post.{i}.boxes"]),
                        "captions": [str(c) for c in f[f"post.{i}.captions"]],
                        "caption_scores": [float(x) for x in f[f"post.{i}.caption_scores"]],
                        "cl_scores": [0.0] * len(f[f"post.{i}.scores"]), "query_id": t(f[f"post.{i}.query_id"]),
                        "vid_duration": torch.tensor(60.0 + 30.0 * i), "pred_seq_len": t(f[f"post.{i}.pred_seq_len"])})
    thr = float(np.median(f["post.0.scores"]))
    batch = E.batch_result_json(results, ["v_a", "v_b"], score_threshold=thr)
    assert list(batch) == ["v_a", "v_b"]
    kept = int((f["post.0.scores"] > thr).sum())
    assert len(batch["v_a"]) == kept > 0
    rec = batch["v_a"][0]
    assert list(rec) == ["timestamp", "raw_box", "label", "proposal_score", "sentence", "sentence_score", "cl_score",
                         "query_id", "vid_duration", "pred_event_count"]
    first = int(np.nonzero(f["post.0.scores"] > thr)[0][0])
    assert rec["timestamp"] == [float(x) for x in f["post.0.boxes"][first]]
    assert rec["sentence"] == str(f["post.0.captions"][first]) and isinstance(rec["proposal_score"], float)
    out = E.new_result_file()
    out["results"].update(batch)
    path = tmp_path / "dvc.json"
    E.save_dvc_json(out, str(path), verbose=True)
    back = json.load(open(path))
    assert back["version"] == "VERSION 1.0" and back["valid_video_num"] == 2
    assert abs(back["avg_proposal_num"] - (len(batch["v_a"]) + len(batch["v_b"])) / 2) < 1e-12


def test_collate_fn_matches_reference():
    """gvl_amd.video_dataset.collate_fn against the reference's collate_fn output on the same synthetic samples
    (tests/golden/collate.npz): every tensor bit for bit, keys and list fields identical."""
    from itertools import chain
    from synth import synth_samples
    from gvl_amd.video_dataset import collate_fn
    f = load("collate")
    dt = collate_fn(synth_samples())
    assert sorted(dt) == [str(k) for k in f["keys"]]
    for k, v in dt.items():
        if isinstance(v, torch.Tensor):
            ref = t(f[k])
            assert v.dtype == ref.dtype and v.shape == ref.shape and torch.equal(v, ref), k
    assert dt["video_key"] == [str(k) for k in f["video_key"]]
    assert np.array_equal(np.array(dt["gt_featstamps"]), f["gt_featstamps"])
    assert list(chain(*dt["cap_raw"])) == [str(c) for c in f["cap_raw"]]
    for i, tg in enumerate(dt["video_target"]):
        assert torch.equal(tg["boxes"], t(f[f"target.{i}.boxes"])) and torch.equal(tg["labels"], t(f[f"target.{i}.labels"]))
        assert tg["masks"] is None and tg["image_id"] == dt["video_key"][i]


def test_oracle_train_step_matches_reference_golden():
    """the oracle's restatement of the TRAINING step (set criterion, Hungarian matching, teacher-forced caption loss,
    autograd backward) against the reference run recorded in pdvc_train.npz: every loss term, the weighted total and the
    gradient norm of every parameter."""
    from oracle import torch_ref as R
    from helpers import pdvc_state, pdvc_dt
    f, g = load("pdvc_eval"), load("pdvc_train")
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in pdvc_state(f).items()}
    dt = pdvc_dt(f)
    dt.update(cap_tensor=t(g["cap_tensor"]), cap_mask=t(g["cap_mask"]))
    losses, total = R.pdvc_train_forward(sd, dt)
    for k in [k for k in g if k.startswith("loss.")]:
        assert abs(float(losses[k[5:]]) - float(g[k])) <= 2e-4 * max(1.0, abs(float(g[k]))), k
    assert abs(float(total) - float(g["final_loss"])) <= 2e-4 * float(g["final_loss"])
    total.backward()
    names = [str(n) for n in g["grad_names"]]
    # tied parameters (pdvc.py:124-140): the reference accumulates both uses into ONE tensor, the flat state dict of the
    # oracle keeps them under two names
    def grad_of(n):
        tot = None
        for k in (n, "transformer.decoder." + n if n.startswith("bbox_head.") else None,
                  "caption_head.0." + n[len("caption_head.1."):] if n.startswith("caption_head.1.") else None,
                  "caption_head.1." + n[len("caption_head.0."):] if n.startswith("caption_head.0.") else None,
                  n[len("transformer.decoder."):] if n.startswith("transformer.decoder.bbox_head.") else None):
            if k is not None and k in sd and sd[k].grad is not None:
                tot = sd[k].grad if tot is None else tot + sd[k].grad
        return tot
    for n, want in zip(names, g["grad_norms"]):
        got = grad_of(n)
        assert got is not None, n
        assert abs(float(got.norm()) - float(want)) <= 2e-3 * max(1.0, float(want)), (n, float(got.norm()), float(want))


@pytest.mark.parametrize("name,seed_w,seed_dt", [("pdvc_anet_full_train_b16", 100, 16), ("pdvc_yc2_train", 512, 4)])
def test_oracle_train_step_at_real_dimensions(name, seed_w, seed_dt):
    """the oracle's training step on the round-4 fixtures at the real model dimensions -- the headline batch of 16 (anet) and the
    long-video configuration (yc2, T = 512, B = 8): every loss term (NaN where the reference's unweighted self-IoU term of a video
    without events is NaN), the weighted total, the matcher's cost through the total, and the float64-accumulated gradient norm of
    every parameter.  Where the fixture carries the reference's own fp32-vs-fp64 error (yc2) it is the yardstick, as on the GPU."""
    from oracle import torch_ref as R
    from helpers import pdvc_state, pdvc_dt
    g = load(name)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in pdvc_state(g, seed=seed_w).items()}
    dt = pdvc_dt(g, feat=int(g["feature_dim"]), seed=seed_dt)
    dt.update(cap_tensor=t(g["cap_tensor"]), cap_mask=t(g["cap_mask"]))
    losses, total = R.pdvc_train_forward(sd, dt)
    for k in [k for k in g if k.startswith("loss.")]:
        want = float(g[k])
        if np.isnan(want):
            assert np.isnan(float(losses[k[5:]])), k
            continue
        assert abs(float(losses[k[5:]]) - want) <= 2e-4 * max(1.0, abs(want)), k
    assert abs(float(total) - float(g["final_loss"])) <= 2e-4 * float(g["final_loss"])
    total.backward()
    names = [str(n) for n in g["grad_names"]]

    def grad_of(n):                                   # tied parameters: see test_oracle_train_step_matches_reference_golden
        tot = None
        for k in (n, "transformer.decoder." + n if n.startswith("bbox_head.") else None,
                  "caption_head.0." + n[len("caption_head.1."):] if n.startswith("caption_head.1.") else None,
                  "caption_head.1." + n[len("caption_head.0."):] if n.startswith("caption_head.0.") else None,
                  n[len("transformer.decoder."):] if n.startswith("transformer.decoder.bbox_head.") else None):
            if k is not None and k in sd and sd[k].grad is not None:
                tot = sd[k].grad if tot is None else tot + sd[k].grad
        return tot
    ref_err = ({n: float(e) / max(1e-3, float(w)) for n, e, w in zip(names, g["grad_norm_f32_err"], g["grad_norms_f64"])}
               if "grad_norm_f32_err" in g else {})
    for n, want in zip(names, g["grad_norms"]):
        got = grad_of(n)
        assert got is not None, n
        tol = max(2e-3, 3.0 * ref_err.get(n, 0.0))
        assert abs(float(got.double().norm()) - float(want)) <= tol * max(1.0, float(want)), (n, float(got.double().norm()), float(want))
