"""GPU parity of the model-level mirror (gvl_amd.pdvc / deformable_transformer / captioner / matcher) against the
golden vectors the imported reference produced with CUDA-op semantics (zero padding in MSDeformAttn, border in the
captioner's MSDeformAttnCap) -- tests/golden/pdvc_eval.npz, captioner_step.npz."""
import os

import numpy as np
import pytest
import torch

from helpers import path_census, load, t, pdvc_state, pdvc_dt, maxerr

pytestmark = pytest.mark.gpu


def to_dev(dt, dev):
    out = dict(dt)
    for k in ("video_tensor", "video_mask", "video_length"):
        out[k] = dt[k].to(dev)
    out["video_target"] = [{k: v.to(dev) for k, v in tg.items()} for tg in dt["video_target"]]
    return out


@pytest.fixture(scope="module")
def built():
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    dev = torch.device("cuda:0")
    f = load("pdvc_eval")
    opt = make_opt(num_queries=8, feature_dim=64, vocab_size=40, max_caption_len=6, device="cuda")
    model, criterion, _, _ = build(opt)
    model.load_state_dict(pdvc_state(f), strict=True)
    model = model.to(dev).eval()
    return f, model, criterion, dev


def test_pdvc_eval_forward_matches_reference(built):
    f, model, criterion, dev = built
    dt = to_dev(pdvc_dt(f), dev)
    with torch.no_grad():
        memory, tshapes, lsi, vr, mflat = model.encode(dt)
        out, loss = model(dt, criterion, None, "queries", eval_mode=True)
    assert maxerr(memory, f["cuda.memory"]) < 2e-4
    assert maxerr(vr, f["cuda.valid_ratios"]) < 1e-6
    assert torch.equal(mflat.cpu(), t(f["cuda.mask_flatten"]))
    assert maxerr(out["event_feat"], f["cuda.hs"][-1]) < 5e-4
    assert maxerr(out["pred_logits"], f["cuda.pred_logits"]) < 5e-4
    assert maxerr(out["pred_boxes"], f["cuda.pred_boxes"]) < 1e-4
    assert maxerr(out["pred_count"], f["cuda.pred_count"]) < 5e-4
    assert maxerr(out["aux_outputs"][0]["pred_boxes"], f["cuda.aux_pred_boxes"]) < 1e-4
    assert maxerr(out["aux_outputs"][0]["pred_logits"], f["cuda.aux_pred_logits"]) < 5e-4
    # greedy captions: token ids identical, log-probs close
    assert torch.equal(out["seq"].cpu(), t(f["cuda.seq"]))
    assert maxerr(out["caption_probs"]["cap_prob_eval"], f["cuda.cap_prob_eval"]) < 5e-4
    # Hungarian index path: bit-exact
    idx, rl = out["matched_indices"]
    for i in range(len(idx)):
        assert torch.equal(torch.stack(idx[i]), t(f[f"cuda.match_{i}"]))
        assert torch.equal(torch.stack(rl[i]), t(f[f"cuda.rl_match_{i}"]))
    for k in ("loss_ce", "loss_counter", "loss_bbox", "loss_giou", "loss_self_iou", "cardinality_error",
              "loss_ce_0", "loss_giou_0"):
        assert maxerr(loss[k].reshape(()), f[f"cuda.loss.{k}"].reshape(())) < 5e-4, k


def test_pdvc_eval_with_gt_proposals_matches_reference(built):
    """transformer_input_type='gt_proposals' (misc/utils.py:32-43; deformable_transformer.py:137-147): the decoder is
    fed the ground-truth segments, refinement is off, box / class / count losses are weighted 0."""
    import copy
    _, model, criterion, dev = built
    f = load("pdvc_gtprop")
    dt = to_dev(pdvc_dt(f), dev)
    dt["gt_boxes"], dt["gt_boxes_mask"] = t(f["gt_boxes"]).to(dev), t(f["gt_boxes_mask"]).to(dev)
    crit = copy.deepcopy(criterion)
    with torch.no_grad():
        out, loss = model(dt, crit, None, "gt_proposals", eval_mode=True)
    assert maxerr(out["pred_boxes"], f["pred_boxes"]) < 1e-5           # = the proposals themselves
    assert maxerr(out["pred_logits"], f["pred_logits"]) < 5e-4
    assert maxerr(out["pred_count"], f["pred_count"]) < 5e-4
    assert maxerr(out["aux_outputs"][0]["pred_boxes"], f["aux_pred_boxes"]) < 1e-5
    assert maxerr(out["aux_outputs"][0]["pred_logits"], f["aux_pred_logits"]) < 5e-4
    assert torch.equal(out["seq"].cpu(), t(f["seq"]))
    assert maxerr(out["caption_probs"]["cap_prob_eval"], f["cap_prob_eval"]) < 5e-4
    for i in range(len(out["matched_indices"][0])):
        assert torch.equal(torch.stack(out["matched_indices"][0][i]), t(f[f"match_{i}"]))
    for k in ("loss_ce", "loss_counter", "loss_bbox", "loss_giou", "loss_self_iou", "loss_ce_0"):
        assert maxerr(loss[k].reshape(()), f[f"loss.{k}"].reshape(())) < 5e-4, k
    names = [str(n) for n in f["weight_names"]]
    assert sorted(crit.weight_dict) == names
    assert [float(crit.weight_dict[k]) for k in names] == [float(v) for v in f["weight_values"]]
    assert crit.matcher.cost_caption == 0


def test_transformer_stages_match_reference(built):
    f, model, criterion, dev = built
    dt = to_dev(pdvc_dt(f), dev)
    with torch.no_grad():
        memory, tshapes, lsi, vr, mflat = model.encode(dt)
        qe = model.query_embed.weight
        init_ref, tgt, ref, qpos = model.transformer.prepare_decoder_input_query(memory, qe)
        pmask = torch.ones(2, qe.shape[0], dtype=torch.bool, device=dev)
        hs, inter = model.transformer.forward_decoder(tgt, ref, memory, tshapes, lsi, vr, qpos, mflat, pmask, False)
    assert maxerr(init_ref, f["cuda.init_reference"]) < 1e-5
    assert maxerr(hs, f["cuda.hs"]) < 5e-4
    assert maxerr(inter, f["cuda.inter_references"]) < 1e-4
    assert tshapes.tolist() == f["tshapes"].tolist() and lsi.tolist() == f["lsi"].tolist()


def test_captioner_single_step_matches_reference(built):
    f, model, criterion, dev = built
    c = load("captioner_step")
    cap = model.caption_head[-1]
    from gvl_amd.deformable_transformer import make_level_tensors
    tshapes, lsi = make_level_tensors(c["tshapes"].tolist(), dev)
    with torch.no_grad():
        logp, (h1, c1) = cap.get_logprobs_state(t(c["it"]).to(dev), (t(c["h0"])[None].to(dev), t(c["c0"])[None].to(dev)),
                                                t(c["hs"]).to(dev), t(c["ref_in"]).to(dev), t(c["memory"]).to(dev),
                                                tshapes, lsi, t(c["mask"]).to(dev))
    assert maxerr(logp, c["logp"]) < 2e-4
    assert maxerr(h1[0], c["h1"]) < 5e-5 and maxerr(c1[0], c["c1"]) < 5e-5


def test_cap_attend_kernel_matches_unfused_reference_order():
    """The fused token-step kernel (offset projection + border sampling + additive attention, with ctx2att pushed
    through the interpolation) against the reference's order of operations restated on the CPU: sample first
    (oracle C sampler, border), THEN ctx2att / tanh / alpha_net / softmax / weighted sum (LSTM_DSA.py:247-266)."""
    import numpy as np
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    from gvl_amd.deformable_transformer import make_level_tensors
    from gvl_amd.ops.modules.ms_deform_attn import temporal_shapes_2d
    from oracle import msda_oracle as O
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    B, Q, C, L, P = 3, 37, 512, 4, 4
    lens = [50, 25, 13, 1]                          # includes a one-row level
    S = sum(lens)
    for RD in (1, 2):
        value = torch.randn(B, S, C, generator=g)
        Wc, bc = torch.randn(C, C, generator=g) / 22, torch.randn(C, generator=g) * 0.1
        Wh, bh = torch.randn(C, C, generator=g) / 22, torch.randn(C, generator=g) * 0.1
        Wo, bo = torch.randn(L * P, 2 * C, generator=g) / 16, torch.randn(L * P, generator=g)
        aw, ab = torch.randn(C, generator=g) / 10, 0.3
        h = torch.randn(B * Q, C, generator=g) * 0.5
        hs = torch.randn(B, Q, C, generator=g)
        ref = torch.rand(B, Q, L, RD, generator=g) * 1.2 - 0.1
        if RD == 2:
            ref[..., 1] = ref[..., 1] * 0.4
        # ---- reference order on the CPU
        off = torch.nn.functional.linear(torch.cat([h.view(B, Q, C), hs], -1), Wo, bo).view(B, Q, 1, L, P)
        T = torch.tensor(lens, dtype=torch.float32)
        if RD == 1:
            x = ref[:, :, None, :, None, 0] + off / T[None, None, None, :, None]
        else:
            x = ref[:, :, None, :, None, 0] + off / P * ref[:, :, None, :, None, 1] * 0.5
        loc = torch.stack((x, torch.full_like(x, 0.5)), -1)
        shapes = np.array([(1, t_) for t_ in lens], np.int64)
        lsi = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        samp = torch.from_numpy(O.msda_sample(value.view(B, S, 1, C).numpy(), shapes, lsi, loc.numpy(), "border"))
        clip = samp.view(B, C, Q, L * P).permute(0, 2, 3, 1).reshape(B * Q, L * P, C)
        att = torch.nn.functional.linear(clip, Wc, bc) + torch.nn.functional.linear(h, Wh, bh)[:, None, :]
        e = torch.tanh(att) @ aw + ab
        alpha = torch.softmax(e, 1)
        want = torch.bmm(alpha.unsqueeze(1), clip).squeeze(1)
        # ---- fused kernel
        slab = torch.cat([value, torch.nn.functional.linear(value, Wc, bc)], -1).contiguous().to(dev)
        tsh, lsi_d = make_level_tensors(lens, dev)
        shapes2d = temporal_shapes_2d(tsh, lsi_d)
        off_hs = torch.nn.functional.linear(hs, Wo[:, C:], bo).contiguous().to(dev)
        got, ga, gl = MSDA.cap_attend(slab, shapes2d, lsi_d, ref.contiguous().to(dev), off_hs, h.to(dev),
                                      Wo[:, :C].contiguous().to(dev),
                                      torch.nn.functional.linear(h, Wh, bh).to(dev), aw.to(dev), ab, L, P, debug=True)
        assert maxerr(gl, x.reshape(B * Q, L * P)) < 1e-4
        assert maxerr(ga, alpha) < 1e-4
        assert maxerr(got, want) < 1e-4
        # the same result left as the fp16 planes + row scale of gvl_gemm_f16x3_f32 (gvl_cap_attend_split_f32)
        pl = MSDA.cap_attend(slab, shapes2d, lsi_d, ref.contiguous().to(dev), off_hs, h.to(dev),
                             Wo[:, :C].contiguous().to(dev), torch.nn.functional.linear(h, Wh, bh).to(dev), aw.to(dev),
                             ab, L, P, planes=True)
        hi_, lo_ = pl.dense()
        back = pl.scale.double()[:, None] * (hi_.double() + lo_.double() / 2048.0)
        rowmax = got.abs().amax(1, keepdim=True).double()
        assert bool(((back - got.double()).abs() <= 2.0 ** -22 * got.abs().double() + 2.0 ** -34 * rowmax).all())
        # ... and from the kernel that keeps the coarse levels' rows in LDS (gvl_cap_attend_split_levels_f32, taken when the
        # caller knows the level starts on the host, L = P = 4 and the rows fit): the same bits
        pl2 = MSDA.cap_attend(slab, shapes2d, lsi_d, ref.contiguous().to(dev), off_hs, h.to(dev),
                              Wo[:, :C].contiguous().to(dev), torch.nn.functional.linear(h, Wh, bh).to(dev), aw.to(dev),
                              ab, L, P, planes=True, host_starts=[int(v) for v in lsi])
        torch.cuda.synchronize()
        assert torch.equal(pl2.hi, pl.hi) and torch.equal(pl2.lo, pl.lo) and torch.equal(pl2.scale, pl.scale)


def test_cap_attend_with_precomputed_offsets_matches_the_kernel_that_multiplies_them():
    """gvl_cap_attend_pre_f32 (h . W_off^T handed in; both halves of levels 2 and 3 -- or, for a longer pyramid, only the ctx2att
    half of level 2 -- in LDS) against gvl_cap_attend_split_f32 on the same inputs: the two differ only in how the 512-term
    offset sum is rounded, which moves a sampling location by ~1e-6 of a row."""
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    from gvl_amd.deformable_transformer import make_level_tensors
    from gvl_amd.ops.modules.ms_deform_attn import temporal_shapes_2d
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    C, L, P = 512, 4, 4
    for B, Q, lens, RD in ((3, 37, [50, 25, 13, 1], 1), (2, 50, [120, 60, 30, 15], 2), (9, 20, [100, 50, 25, 13], 1)):
        S = sum(lens)
        lsi = [0] + list(np.cumsum(lens)[:-1])
        assert MSDA.cap_attend_pre_applicable(S, L, P, lsi)
        slab = torch.randn(B, S, 2 * C, generator=g).to(dev)
        Wo = (torch.randn(L * P, C, generator=g) / 16).to(dev)
        h = (torch.randn(B * Q, C, generator=g) * 0.5).to(dev)
        off_hs = torch.randn(B, Q, L * P, generator=g).to(dev)
        ref = torch.rand(B, Q, L, RD, generator=g) * 1.2 - 0.1
        if RD == 2:
            ref[..., 1] = ref[..., 1] * 0.4
        ref = ref.contiguous().to(dev)
        att_h = torch.randn(B * Q, C, generator=g).to(dev)
        aw, ab = (torch.randn(C, generator=g) / 10).to(dev), 0.3
        tsh, lsi_d = make_level_tensors(lens, dev)
        shapes2d = temporal_shapes_2d(tsh, lsi_d)
        want = MSDA.cap_attend(slab, shapes2d, lsi_d, ref, off_hs, h, Wo, att_h, aw, ab, L, P, planes=True)
        # the operands as the token loop hands them over: columns of ONE wider matrix [h2att(h) | h W_off^T]
        wide = torch.cat([att_h, (h.double() @ Wo.double().t()).float()], 1).contiguous()
        got = MSDA.cap_attend_pre(slab, shapes2d, lsi_d, ref, off_hs, wide[:, C:], wide[:, :C], aw, ab, L, P, lsi)
        torch.cuda.synchronize()

        def dense(pl):
            hi_, lo_ = pl.dense()
            return pl.scale.double()[:, None] * (hi_.double() + lo_.double() / 2048.0)
        assert maxerr(dense(got), dense(want)) < 2e-5
    assert not MSDA.cap_attend_pre_applicable(375, L, P, [0, 200, 300, 350])       # levels 2 + 3 + 3 = 100 rows: no LDS form
    assert not MSDA.cap_attend_pre_applicable(S, L, P, None)


def test_row_argmax_lse_matches_torch():
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(777, 8518, generator=g) * 3
    x[5, 100] = x[5, 4000] = 50.0              # tie: the first maximal index wins (torch.max semantics)
    x[6] = 1.0                                  # constant row
    xs = torch.randn(9, 41, generator=g)        # fewer columns than threads
    i2, l2 = MSDA.row_argmax_lse(xs.to(dev))
    w2, wi2 = torch.max(torch.log_softmax(xs.double(), 1), 1)
    assert torch.equal(i2.cpu(), wi2) and maxerr(l2, w2.float()) < 1e-5
    idx, lp = MSDA.row_argmax_lse(x.to(dev))
    want_lp, want_idx = torch.max(torch.log_softmax(x.double(), 1), 1)
    assert torch.equal(idx.cpu()[:5], want_idx[:5]) and int(idx[5]) == 100 and int(idx[6]) == 0
    assert torch.equal(idx.cpu()[7:], want_idx[7:])
    assert maxerr(lp, want_lp.float()) < 1e-4


def test_sample_function_backward_matches_oracle_autograd():
    """MSDASampleFunction (HIP sampler + its backward) vs autograd through the oracle's grid_sample restatement."""
    from gvl_amd.ops.functions import MSDASampleFunction
    from oracle import torch_ref as R
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    B, M, D, Q, L, P = 2, 1, 96, 5, 3, 4
    lens = [9, 5, 1]
    S = sum(lens)
    value = torch.randn(B, S, M, D, generator=g)
    loc = torch.rand(B, Q, M, L, P, 2, generator=g) * 1.4 - 0.2
    loc[..., 1] = 0.5
    shapes = torch.tensor([(1, t_) for t_ in lens])
    lsi = torch.tensor([0, 9, 14])
    gs = torch.randn(B * M, D, Q, L, P, generator=g)
    v1, l1 = value.clone().requires_grad_(), loc.clone().requires_grad_()
    R.msda_core(v1, shapes, l1, None, "border", return_value=True).backward(gs)
    v2, l2 = value.to(dev).requires_grad_(), loc.to(dev).requires_grad_()
    out = MSDASampleFunction.apply(v2, shapes.to(dev), lsi.to(dev), l2, "border")
    out.backward(gs.to(dev))
    assert maxerr(v2.grad, v1.grad) < 1e-4
    assert maxerr(l2.grad, l1.grad) < 1e-3


def test_pdvc_train_step_matches_reference():
    """One training forward/backward (set losses + Hungarian matcher + teacher-forced captioner on the matched
    queries) against the reference run with all dropout = 0: every loss term, the matcher indices, and the gradient
    norm of every parameter; selected gradients element-wise."""
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    dev = torch.device("cuda:0")
    f = load("pdvc_eval")
    g = load("pdvc_train")
    opt = make_opt(num_queries=8, feature_dim=64, vocab_size=40, max_caption_len=6, device="cuda",
                   transformer_dropout_prob=0.0, drop_prob=0.0)
    model, criterion, _, _ = build(opt)
    model.load_state_dict(pdvc_state(f), strict=True)
    model = model.to(dev).train()
    dt = to_dev(pdvc_dt(f), dev)
    dt.update(cap_tensor=t(g["cap_tensor"]).to(dev), cap_mask=t(g["cap_mask"]).to(dev),
              gt_boxes_mask=torch.tensor([[1, 1, 1], [1, 1, 0]], dtype=torch.bool, device=dev))
    out, loss = model(dt, criterion, None, "queries")
    wd = criterion.weight_dict
    final = sum(loss[k] * wd[k] for k in loss.keys() if k in wd)
    final.backward()
    for k in [k for k in g if k.startswith("loss.")]:
        assert maxerr(loss[k[5:]].reshape(()), g[k].reshape(())) < 1e-3, k
    assert abs(float(final) - float(g["final_loss"])) < 2e-3
    for i, (a, b) in enumerate(out["matched_indices"][0]):
        assert torch.equal(torch.stack([a, b]), t(g[f"match_{i}"]))
    params = dict(model.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    assert sorted(n for n, p_ in params.items() if p_.grad is not None) == names
    for n, want in zip(names, g["grad_norms"]):
        got = float(params[n].grad.norm())
        assert abs(got - float(want)) <= 2e-3 * max(1.0, float(want)), (n, got, float(want))
    for k in [k for k in g if k.startswith("grad.")]:
        want = g[k]
        assert maxerr(params[k[5:]].grad, want) <= 2e-3 * max(1.0, float(abs(want).max())), k


def test_caption_cost_in_the_matcher_fails_as_the_reference_does():
    """set_cost_caption > 0 routes training through parallel_prediction_full_train (pdvc.py:305-309, :322-432).  With the
    LSTM-DSA captioner the REFERENCE fails inside it (caption_prediction iterates indices=None, pdvc.py:743): the probe
    tests/golden/make_golden.py:make_full_train_probe ran the reference and recorded the exception
    (tests/golden/full_train_probe.npz).  There is no output to mirror; gvl_amd raises the same exception type and message
    at the same point of the forward (after the decoder), and the evaluation forward -- which never takes that branch in
    the reference either (pdvc.py:301-304) -- is unaffected by the option."""
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    probe = load("full_train_probe")
    assert bool(probe["raised"]) and str(probe["exc_type"]) == "TypeError" and "pdvc.py:743" in str(probe["where"])
    dev = torch.device("cuda:0")
    f, g = load("pdvc_eval"), load("pdvc_train")
    opt = make_opt(num_queries=8, feature_dim=64, vocab_size=40, max_caption_len=6, device="cuda", set_cost_caption=1.0)
    model, criterion, _, _ = build(opt)
    model.load_state_dict(pdvc_state(f), strict=True)
    model = model.to(dev).train()
    dt = to_dev(pdvc_dt(f), dev)
    dt.update(cap_tensor=t(g["cap_tensor"]).to(dev), cap_mask=t(g["cap_mask"]).to(dev),
              gt_boxes_mask=torch.tensor([[1, 1, 1], [1, 1, 0]], dtype=torch.bool, device=dev))
    with pytest.raises(TypeError) as e:
        model(dt, criterion, None, "queries")
    # the reference's own exception type and message, followed by the explanation a user needs
    assert str(e.value).startswith(str(probe["exc_message"])) and "set_cost_caption" in str(e.value)
    model.eval()
    with torch.no_grad():
        out, loss = model(dt, criterion, None, "queries", eval_mode=True)
    assert maxerr(out["pred_boxes"], f["cuda.pred_boxes"]) <= 2e-4



def test_token_loop_gemm_paths_agree(built, monkeypatch):
    """the captioner's token loop on gvl_gemm_f16x3 (default; argmax fused, cell / attention kernels emitting planes) and
    on the fp32 library GEMMs (GVL_GEMM=f32): the same greedy tokens, log-probabilities within fp32 noise"""
    f, model, criterion, dev = built
    dt = to_dev(pdvc_dt(f), dev)
    cap = model.caption_head[-1]
    outs = {}
    with torch.no_grad():
        cap.graph_decode = False
        for mode in ("f16x3", "f32"):
            if mode == "f32":
                monkeypatch.setenv("GVL_GEMM", "f32")
            else:
                monkeypatch.delenv("GVL_GEMM", raising=False)
            outs[mode], _ = model(dt, None, None, "queries", eval_mode=True)
    monkeypatch.delenv("GVL_GEMM", raising=False)
    assert torch.equal(outs["f16x3"]["seq"], outs["f32"]["seq"])
    assert maxerr(outs["f16x3"]["caption_probs"]["cap_prob_eval"], outs["f32"]["caption_probs"]["cap_prob_eval"]) < 2e-5


def test_graph_replayed_decoding_equals_eager(built):
    """The hipGraph replay of the greedy decoding loop returns exactly what the eager loop returns, also when the
    inputs change between replays (static input buffers are refreshed)."""
    f, model, criterion, dev = built
    dt = to_dev(pdvc_dt(f), dev)
    cap = model.caption_head[-1]
    with torch.no_grad():
        cap.graph_decode = False
        out_e, _ = model(dt, None, None, "queries", eval_mode=True)
        cap.graph_decode = True
        try:
            out_g, _ = model(dt, None, None, "queries", eval_mode=True)
            assert torch.equal(out_g["seq"], out_e["seq"])
            assert maxerr(out_g["caption_probs"]["cap_prob_eval"], out_e["caption_probs"]["cap_prob_eval"]) < 1e-5
            dt2 = dict(dt)
            dt2["video_tensor"] = dt["video_tensor"].flip(0).contiguous()
            dt2["video_mask"] = dt["video_mask"].flip(0).contiguous()
            out_g2, _ = model(dt2, None, None, "queries", eval_mode=True)       # replay with new inputs
            cap.graph_decode = False
            out_e2, _ = model(dt2, None, None, "queries", eval_mode=True)
            assert torch.equal(out_g2["seq"], out_e2["seq"])
            assert maxerr(out_g2["caption_probs"]["cap_prob_eval"], out_e2["caption_probs"]["cap_prob_eval"]) < 1e-5
        finally:
            cap.graph_decode = False


@pytest.mark.parametrize("split", [False, True])
def test_graphed_train_step_equals_eager(split):
    """GraphedTrainStep (whole step replayed from a hipGraph, Hungarian matching on the device) computes what the eager
    TrainStep computes.  Adam's first updates are lr * sign(g) for near-zero gradients, i.e. ill-conditioned under
    1e-7 gradient noise, so the comparison is made with lr = 1e-10 (parameters stay identical on both sides; Adam(capturable) divides by lr) on what the
    step produces: losses, gradients and Adam's moment estimates, over several replays with alternating batches."""
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    from gvl_amd.parallel import TrainStep, GraphedTrainStep
    dev = torch.device("cuda:0")
    f = load("pdvc_eval")
    g = load("pdvc_train")
    opt = make_opt(num_queries=8, feature_dim=64, vocab_size=40, max_caption_len=6, device="cuda",
                   transformer_dropout_prob=0.0, drop_prob=0.0, lr=1e-10, weight_decay=0.0)

    def mk():
        m, c, _, _ = build(opt)
        m.load_state_dict(pdvc_state(f), strict=True)
        return m.to(dev).train(), c

    def batch(flip):
        dt = to_dev(pdvc_dt(f), dev)
        dt.update(cap_tensor=t(g["cap_tensor"]).to(dev), cap_mask=t(g["cap_mask"]).to(dev),
                  gt_boxes_mask=torch.tensor([[1, 1, 1], [1, 1, 0]], dtype=torch.bool, device=dev))
        if flip:                                   # a second batch with the same layout but different features
            dt["video_tensor"] = dt["video_tensor"] * 0.5 + 0.1
        return dt

    (model_a, crit_a), (model_b, crit_b) = mk(), mk()
    eager = TrainStep(model_a, crit_a, opt, capturable=True)
    # split=True: the data-parallel form (forward/backward graph, eager gradient exchange, clip/Adam graph), exercised
    # here on one process where the exchange is the identity
    graphed = GraphedTrainStep(model_b, crit_b, opt, warmup=1, split_exchange=split)
    seen = []                # (warm-up and capture are side-effect free: both sides make exactly one update per batch)
    for step in range(5):
        la, _ = eager(batch(step % 2 == 1))
        lb, _ = graphed(batch(step % 2 == 1))      # step 0 captures then replays, later steps only replay
        assert abs(float(la) - float(lb)) < 1e-4 * max(1.0, abs(float(la))), (step, float(la), float(lb))
        seen.append(float(lb))
        for (n, pa), pb in zip(model_a.named_parameters(), model_b.parameters()):
            # dead parameters (e.g. the captioner's unused attention_weights / output_proj) get no gradient: None with
            # hand-over gradients, zeros in the flat buffer of the data-parallel form
            if pa.grad is None:
                assert pb.grad is None or float(pb.grad.abs().max()) == 0.0, (step, n)
                continue
            assert maxerr(pa.grad, pb.grad) <= 1e-4 * max(1.0, float(pa.grad.abs().max())), (step, n)
    assert abs(seen[0] - float(g["final_loss"])) < 2e-3 and abs(seen[0] - seen[2]) < 1e-4 and abs(seen[0] - seen[1]) > 1e-3
    sa, sb = eager.optimizer.state, graphed.optimizer.state
    for pa, pb in zip(eager.params, graphed.params):
        if pa.grad is None:
            continue
        assert float(sa[pa]["step"]) == float(sb[pb]["step"]) == 5.0
        assert maxerr(sa[pa]["exp_avg"], sb[pb]["exp_avg"]) <= 1e-4 * max(1.0, float(sa[pa]["exp_avg"].abs().max()))


def test_graphed_train_step_dropout_advances_rng():
    """With dropout on, successive replays of the captured step must draw new masks (PyTorch's graph-safe Philox
    offsets): the loss on the same batch differs from replay to replay, and stays finite."""
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    from gvl_amd.parallel import GraphedTrainStep
    dev = torch.device("cuda:0")
    f = load("pdvc_eval")
    g = load("pdvc_train")
    opt = make_opt(num_queries=8, feature_dim=64, vocab_size=40, max_caption_len=6, device="cuda", lr=1e-10,
                   weight_decay=0.0)                      # default dropouts: transformer 0.1, captioner 0.5
    model, crit, _, _ = build(opt)
    model.load_state_dict(pdvc_state(f), strict=True)
    model = model.to(dev).train()
    dt = to_dev(pdvc_dt(f), dev)
    dt.update(cap_tensor=t(g["cap_tensor"]).to(dev), cap_mask=t(g["cap_mask"]).to(dev),
              gt_boxes_mask=torch.tensor([[1, 1, 1], [1, 1, 0]], dtype=torch.bool, device=dev))
    step = GraphedTrainStep(model, crit, opt, warmup=1)
    losses = [float(step(dt)[0]) for _ in range(4)]
    assert all(np.isfinite(losses)) and len({round(x, 5) for x in losses}) == 4, losses


# ---- BASELINE.json config 4: cfgs/yc2_tsn_dvc.yml at its real dimensions on long videos (T = 512) ----------------

@pytest.fixture(scope="module")
def built_yc2():
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    dev = torch.device("cuda:0")
    f = load("pdvc_yc2")
    opt = make_opt("yc2_tsn_dvc", max_caption_len=8, frame_embedding_num=512, device="cuda")
    assert (opt.feature_dim, opt.num_queries, opt.vocab_size) == (int(f["feature_dim"]), int(f["num_queries"]),
                                                                  int(f["vocab_size"]))
    model, criterion, _, _ = build(opt)
    model.load_state_dict(pdvc_state(f, seed=512), strict=True)
    return f, model.to(dev).eval(), criterion, dev


def _check_yc2(f, out, loss, memory, tol, seq_exact):
    """tol bounds the encoder memory (relative to its scale) and, x5, the sigmoid box heads; logits / counts / decoder
    features get 25x: at T = 512 the decoder's sampling amplifies position noise (d sample / d loc = T_l * dv), so
    even the reference's own fp32 run sits 1.4e-4 (boxes) / 1.3e-3 (logits) away from an fp64 evaluation of the same
    model, and gvl_amd's fp32 run 3.1e-4 / 2.6e-3 (tools/stage_times.py-style probe recorded in DESIGN_LOG.md section 5)."""
    ms = float(np.abs(f["memory_rows"]).max())
    assert maxerr(memory[:, ::8], f["memory_rows"]) <= tol * max(1.0, ms)
    assert abs(float(memory.double().sum()) - float(f["memory_sum"])) <= tol * 960 * 2 * 512 ** 0.5
    assert maxerr(out["pred_boxes"], f["pred_boxes"]) <= 5 * tol
    assert maxerr(out["aux_outputs"][0]["pred_boxes"], f["aux_pred_boxes"]) <= 5 * tol
    assert maxerr(out["pred_logits"], f["pred_logits"]) <= 25 * tol
    assert maxerr(out["pred_count"], f["pred_count"]) <= 25 * tol
    assert maxerr(out["event_feat"][:, ::4], f["event_feat"]) <= 25 * tol * max(1.0, float(np.abs(f["event_feat"]).max()))
    # greedy captions: with random weights the top-2 logit gap is below the noise floor above for ~1 % of the tokens
    # and a flipped token changes the rest of its row, so rows are compared where the token sequences agree
    seq, ref_seq = out["seq"].cpu(), t(f["seq"])
    seq, ref_seq = seq.reshape(-1, seq.shape[-1]), ref_seq.reshape(-1, ref_seq.shape[-1])
    n = min(seq.shape[1], ref_seq.shape[1])
    row_same = (seq[:, :n] == ref_seq[:, :n]).all(1)
    same = (seq[:, :n] == ref_seq[:, :n]).float().mean()
    lp = out["caption_probs"]["cap_prob_eval"].float().cpu()
    lp, ref_lp = lp.reshape(-1, lp.shape[-1])[:, :n], t(f["cap_prob_eval"]).reshape(-1, ref_seq.shape[-1])[:, :n]
    assert maxerr(lp[row_same], ref_lp[row_same]) <= 25 * tol
    if seq_exact:
        assert float(same) >= 0.95 and float(row_same.float().mean()) >= 0.9
        for i in range(len(out["matched_indices"][0])):
            assert torch.equal(torch.stack(out["matched_indices"][0][i]), t(f[f"match_{i}"]))
    for k in ("loss_ce", "loss_giou", "loss_counter", "loss_self_iou"):
        assert maxerr(loss[k].reshape(()), f[f"loss.{k}"].reshape(())) <= 25 * tol, k
    return float(same)


def test_yc2_long_video_eval_matches_reference(built_yc2):
    """fp32: 3072-d input, S = 960 (> 639 rows: level 0 of the value slab stays in global memory, levels 1..3 in LDS)
    through base encoder, fused encoder / decoder attention, captioner and matcher -- against the reference run."""
    from gvl_amd import _lib
    f, model, criterion, dev = built_yc2
    dt = to_dev(pdvc_dt(f, feat=int(f["feature_dim"]), seed=4), dev)
    with torch.no_grad():
        memory = model.encode(dt)[0]
        assert _lib.lib().gvl_msda_last_impl() == 3                # the fused temporal kernels served S = 960
        out, loss = model(dt, criterion, None, "queries", eval_mode=True)
    _check_yc2(f, out, loss, memory, 2e-4, seq_exact=True)


@pytest.mark.parametrize("policy", ["bf16", "fp32", "f16"])
def test_yc2_long_video_eval_under_bf16_autocast(built_yc2, policy, monkeypatch):
    """policy = "f16" (the default, GVL_AUTOCAST_INFERENCE unset): an INFERENCE forward under torch.autocast stays on the
    hand-written fp32-storage path and spends ONE fp16 matrix-core product per fp32 product (operands rounded to 11 bits at
    their row scale; gvl_amd/pdvc.py) -- held to the fp32 golden at 10x the fp32 tolerances, far inside what the bf16 path
    below can be held to.  policy = "fp32" (GVL_AUTOCAST_INFERENCE=fp32): the same path with the exact products, an fp32
    island: reproduces the fp32 golden at the fp32 tolerances.  policy = "bf16" (GVL_AUTOCAST_INFERENCE=bf16), the bf16-storage path:
    The same forward under torch.autocast(bfloat16) (BASELINE config 4 names bf16): GEMMs on bf16 MFMA, the
    deformable attention on bf16 storage with fp32 locations, the decoding loop on the bf16-input token-step kernels.
    What can be pinned at model level: the encoder memory (two layers of rounded GEMMs: mean error < 0.5 % of its
    scale, max < 2 %).  Behind the decoder nothing tighter than statistics is meaningful with RANDOM weights: the
    sampling offsets come out of a bf16 GEMM (|off| of several frames -> ~0.1 frame of rounding at T = 512) and
    d sample / d loc = T_l * dv, so box / logit errors of 0.03 / 0.2 (median) are the noise of the number format,
    not of the kernels -- those are pinned bit-for-bit against the fp32 kernels in test_gpu_bf16.py."""
    f, model, criterion, dev = built_yc2
    monkeypatch.setenv("GVL_AUTOCAST_INFERENCE", "" if policy == "f16" else policy)
    dt = to_dev(pdvc_dt(f, feat=int(f["feature_dim"]), seed=4), dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        memory = model.encode(dt)[0]
        out, loss = model(dt, criterion, None, "queries", eval_mode=True)
    if policy != "bf16":
        from gvl_amd import MultiScaleDeformableAttention as MSDA
        assert out["pred_boxes"].dtype == torch.float32
        with torch.no_grad(), MSDA.f16_products(1 if policy == "f16" else 3):
            memory32 = model.encode(dt)[0]                   # (encode() on its own is not the island: run it the same way)
        if policy == "fp32":
            _check_yc2(f, out, loss, memory32, 2e-4, seq_exact=True)
        else:
            # against the exact forward of the same model.  At T = 512 the decoder's sampling amplifies position noise
            # (d sample / d loc = T_l * dv, see _check_yc2), so single boxes move by 1e-2 where the mean moves by 1e-3; the
            # bf16 policy below is held to a MEAN box error of 0.08
            ms = float(np.abs(f["memory_rows"]).max())
            assert maxerr(memory32[:, ::8], f["memory_rows"]) <= 4e-3 * max(1.0, ms)
            with torch.no_grad():
                exact, _ = model(dt, criterion, None, "queries", eval_mode=True)
            monkeypatch.setenv("GVL_AUTOCAST_INFERENCE", "bf16")
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                lowp, _ = model(dt, criterion, None, "queries", eval_mode=True)
            stats, stats_bf = {}, {}
            for k in ("pred_boxes", "pred_logits", "pred_count"):
                d, d_bf = (out[k].float() - exact[k].float()).abs(), (lowp[k].float() - exact[k].float()).abs()
                stats[k], stats_bf[k] = (float(d.mean()), float(d.max())), (float(d_bf.mean()), float(d_bf.max()))
            n = min(out["seq"].shape[-1], exact["seq"].shape[-1])
            stats["tokens"] = float((out["seq"][..., :n] == exact["seq"][..., :n]).float().mean())
            print("f16 policy vs exact:", stats, "| bf16 policy vs exact:", stats_bf)
            assert 0 < stats["pred_logits"][1]                                  # it IS the reduced product
            # measured 1.6e-3 / 1.3e-2 (boxes mean / max), 1.5e-2 (logits mean), 0.72 (tokens equal; random-like weights)
            assert stats["pred_boxes"][0] <= 4e-3 and stats["pred_boxes"][1] <= 5e-2
            assert stats["pred_logits"][0] <= 4e-2 and stats["pred_count"][0] <= 4e-2
            assert stats["tokens"] >= 0.6
            for k in ("pred_boxes", "pred_logits", "pred_count"):              # ... and closer than the bf16-storage path
                assert stats[k][0] <= 0.5 * stats_bf[k][0], (k, stats[k], stats_bf[k])
            assert all(torch.isfinite(v.float()).all() for v in loss.values())
        return
    ms = float(np.abs(f["memory_rows"]).max())
    d = (memory[:, ::8].float().cpu() - t(f["memory_rows"])).abs()
    assert float(d.max()) <= 2e-2 * ms and float(d.mean()) <= 5e-3 * ms
    for k in ("pred_boxes", "pred_logits", "pred_count"):
        assert torch.isfinite(out[k].float()).all()
    assert float((out["pred_boxes"].float().cpu() - t(f["pred_boxes"])).abs().mean()) <= 0.08
    assert out["seq"].shape[:2] == (2, int(f["num_queries"]))
    assert all(torch.isfinite(v.float()).all() for v in loss.values())


@pytest.mark.parametrize("ref_dim", [2, 1])
def test_teacher_forced_loop_fused_equals_step_by_step(built, ref_dim, monkeypatch):
    """TeacherForcedLoop (k_cap_train_* / k_lstm_train_* + deferred weight-gradient GEMMs) against the step-by-step
    formulation built from the sampler op + PyTorch autograd (itself pinned to the reference by the train golden):
    log-probs and the gradients of every captioner parameter and of hs / reference / memory."""
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    f, model, criterion, dev = built
    cap = model.caption_head[-1]
    torch.manual_seed(5)
    B, Qm, steps = 2, 3, 5
    dt = to_dev(pdvc_dt(f), dev)
    with torch.no_grad():
        memory, tshapes, lsi, vr, mflat = model.encode(dt)
    memory = (memory + 0.05 * torch.randn_like(memory)).requires_grad_()
    hs = torch.randn(B, Qm, 512, device=dev, requires_grad=True)
    ref = (torch.rand(B, Qm, ref_dim, device=dev) * (0.5 if ref_dim == 2 else 1.0) + 0.1).requires_grad_()
    others = {"memory": memory, "mask_flatten": mflat, "spatial_shapes": tshapes, "level_start_index": lsi,
              "valid_ratios": vr}
    seq = torch.randint(1, 40, (B * Qm, steps + 1), device=dev)
    seq[:, 0] = 0
    gout = None
    res = {}
    params = dict(cap.named_parameters())
    for mode in (True, False):
        cap.core.fused_train = mode
        for p_ in list(params.values()) + [memory, hs, ref]:
            p_.grad = None
        MSDA.profile_enable(True)
        out = cap(hs, ref, others, seq, steps=steps)
        if gout is None:
            gout = torch.randn_like(out)
        (out * gout).sum().backward()
        torch.cuda.synchronize()
        MSDA.profile_enable(False)
        tags = [e[0] for e in MSDA.profile_collect()]
        assert tags.count("cap_train_fwd") == (steps if mode else 0) and tags.count("cap_train_bwd") == (steps if mode else 0)
        assert tags.count("lstm_train") == (2 * steps if mode else 0)
        res[mode] = (out.detach().clone(), {k: (v.grad.clone() if v.grad is not None else None)
                                            for k, v in list(params.items()) + [("memory", memory), ("hs", hs), ("ref", ref)]})
    cap.core.fused_train = True
    assert maxerr(res[True][0], res[False][0]) < 2e-4
    for k, g_ref in res[False][1].items():
        g = res[True][1][k]
        if g_ref is None:
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        assert g is not None, k
        scale_ = max(1e-3, float(g_ref.abs().max()))
        assert float((g - g_ref).abs().max()) <= 2e-3 * scale_, (k, float((g - g_ref).abs().max()), scale_)


def test_graphed_eval_forward_equals_eager(built):
    """GraphedEvalForward (whole eval forward replayed from one hipGraph, caption tensor trimmed after the replay) returns
    what the eager forward returns, also on a second batch of the same layout and again on the first one."""
    from gvl_amd.parallel import GraphedEvalForward
    f, model, criterion, dev = built
    g = GraphedEvalForward(model, criterion)
    dts = [to_dev(pdvc_dt(f), dev), to_dev(pdvc_dt(f), dev)]
    dts[1]["video_tensor"] = dts[1]["video_tensor"] * 0.7 + 0.05
    for k in (0, 1, 0):
        with torch.no_grad():
            ref_out, ref_loss = model(dts[k], criterion, None, "queries", eval_mode=True)
        out, loss = g(dts[k])
        assert out["seq"].shape == ref_out["seq"].shape and torch.equal(out["seq"], ref_out["seq"])
        assert maxerr(out["caption_probs"]["cap_prob_eval"], ref_out["caption_probs"]["cap_prob_eval"]) < 1e-5
        for key in ("pred_logits", "pred_boxes", "pred_count"):
            assert maxerr(out[key], ref_out[key]) < 1e-5, key
        for key, v in ref_loss.items():
            assert maxerr(loss[key].reshape(()), v.reshape(())) < 1e-5 or (v != v).all(), key
        for a_, b_ in zip(out["matched_indices"][0], ref_out["matched_indices"][0]):
            assert torch.equal(a_[0], b_[0]) and torch.equal(a_[1], b_[1])
    assert len(g.graphs) == 1
    # eval golden through the graph as well
    out, loss = g(dts[0])
    assert torch.equal(out["seq"].cpu(), t(f["cuda.seq"]))
    assert maxerr(out["pred_boxes"], f["cuda.pred_boxes"]) < 1e-4


# ---- BASELINE.json configs 1-2 at the real model dimensions (300 queries, vocabulary 8517, 30 tokens) ------------

def test_anet_full_dimension_eval_matches_reference():
    """cfgs/anet_tsp_ssvg.yml at its real dimensions, eager and through GraphedEvalForward, against the reference run:
    encoder memory, heads, matched indices, losses; greedy captions row by row (a near-tie flip changes the rest of
    that row, so rows are compared where the token sequences agree; >= 95 % of the tokens must agree)."""
    from gvl_amd.config import make_opt
    from gvl_amd.parallel import GraphedEvalForward
    from gvl_amd.pdvc import build
    dev = torch.device("cuda:0")
    f = load("pdvc_anet_full")
    opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
    assert (opt.feature_dim, opt.vocab_size, opt.max_caption_len) == (int(f["feature_dim"]), int(f["vocab_size"]),
                                                                      int(f["max_caption_len"]))
    model, criterion, _, _ = build(opt)
    model.load_state_dict(pdvc_state(f, seed=100), strict=True)
    model = model.to(dev).eval()
    dt = to_dev(pdvc_dt(f, feat=int(f["feature_dim"]), seed=6), dev)
    graphed = GraphedEvalForward(model, criterion)
    for mode in ("eager", "graph"):
        with torch.no_grad():
            memory = model.encode(dt)[0]
            out, loss = model(dt, criterion, None, "queries", eval_mode=True) if mode == "eager" else graphed(dt)
        ms = float(np.abs(f["memory_rows"]).max())
        assert maxerr(memory[:, ::4], f["memory_rows"]) <= 2e-4 * max(1.0, ms)
        assert maxerr(out["pred_boxes"], f["pred_boxes"]) <= 2e-4
        assert maxerr(out["aux_outputs"][0]["pred_boxes"], f["aux_pred_boxes"]) <= 2e-4
        # (the noise floor of an fp32 evaluation of this model -- the reference's own fp32 run against its fp64 run -- is
        #  5e-5 / 6e-6 / 3e-5 for logits / boxes / counts: tests/golden/pdvc_anet_full_f64.npz, asserted against in
        #  tests/test_gpu_full_dims.py; two fp32 runs may differ by the sum of their errors)
        assert maxerr(out["pred_logits"], f["pred_logits"]) <= 3e-4
        assert maxerr(out["pred_count"], f["pred_count"]) <= 3e-4
        assert maxerr(out["event_feat"][:, ::8], f["event_feat"]) <= 1e-3 * max(1.0, float(np.abs(f["event_feat"]).max()))
        for i in range(len(out["matched_indices"][0])):
            assert torch.equal(torch.stack(out["matched_indices"][0][i]), t(f[f"match_{i}"]))
        for k in ("loss_ce", "loss_giou", "loss_counter", "loss_self_iou"):
            assert maxerr(loss[k].reshape(()), f[f"loss.{k}"].reshape(())) <= 1e-3, k
        seq, ref_seq = out["seq"].cpu(), t(f["seq"])
        assert seq.shape == ref_seq.shape == (2, 300, int(f["max_caption_len"]))
        seq, ref_seq = seq.reshape(-1, seq.shape[-1]), ref_seq.reshape(-1, ref_seq.shape[-1])
        row_same = (seq == ref_seq).all(1)
        # greedy tokens: equal to the reference's except where two logits tie to within the fp32 noise of the products in
        # front of them (a flipped tie changes that token and the rest of its caption): at most 3 of the 600 captions
        assert float((seq == ref_seq).float().mean()) >= 0.999 and int((~row_same).sum()) <= 3
        lp = out["caption_probs"]["cap_prob_eval"].float().cpu().reshape(-1, seq.shape[-1])
        assert maxerr(lp[row_same], t(f["cap_prob_eval"]).reshape(-1, seq.shape[-1])[row_same]) <= 2e-3


def test_headline_batch_of_16_eval_matches_reference():
    """BASELINE.json config 1 as the bench runs it -- B = 16 videos, T = 100, 300 queries, 30 caption tokens -- against the
    reference's run of the same batch (tests/golden/pdvc_anet_full_b16.npz; VERDICT r3 weak 1a: the model-level goldens were
    B = 2).  Eager and through GraphedEvalForward (the timed path): heads, counts, refined boxes of both layers, matched
    indices incl. event-less videos, losses, greedy tokens and their log-probabilities."""
    from gvl_amd.config import make_opt
    from gvl_amd.parallel import GraphedEvalForward
    from gvl_amd.pdvc import build
    dev = torch.device("cuda:0")
    f = load("pdvc_anet_full_b16")
    opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
    model, criterion, _, _ = build(opt)
    model.load_state_dict(pdvc_state(f, seed=100), strict=True)
    model = model.to(dev).eval()
    dt = to_dev(pdvc_dt(f, feat=int(f["feature_dim"]), seed=16), dev)
    assert dt["video_tensor"].shape[0] == 16
    graphed = GraphedEvalForward(model, criterion)
    for mode in ("eager", "graph"):
        with torch.no_grad():
            if mode == "eager":
                (out, loss), ran = path_census(lambda: model(dt, criterion, None, "queries", eval_mode=True))
                # the goldens below would also pass on the PyTorch formulation of the layers: what SERVED the call is asserted --
                # base encoder + 2 encoder + 2 decoder layers + heads on gvl_linear_f16x3_f32 / the LayerNorm / attention-core /
                # geometry kernels, four fused deformable-attention launches, the token loop's split-fp16 products
                assert ran["layer_gemm"] >= 30 and ran["layer_norm_etc"] >= 10 and ran["fwd_t1d_d64"] == 4, ran
                # (per token: the gate product + cell and the vocabulary product under the gemm_f16x3 tag; h2att(h) rides in the
                #  greedy reduction's launch, tagged row_argmax_lse)
                assert ran["gemm_f16x3"] >= 2 * 30 and ran["cap_attend"] >= 30 and ran["row_argmax_lse"] >= 30, ran
            else:
                out, loss = graphed(dt)
        assert maxerr(out["pred_boxes"], f["pred_boxes"]) <= 2e-4
        assert maxerr(out["aux_outputs"][0]["pred_boxes"], f["aux_pred_boxes"]) <= 2e-4
        assert maxerr(out["pred_logits"], f["pred_logits"]) <= 3e-4
        assert maxerr(out["pred_count"], f["pred_count"]) <= 3e-4
        assert maxerr(out["event_feat"][:, ::8, ::4], f["event_feat"]) <= 1e-3 * max(1.0, float(np.abs(f["event_feat"]).max()))
        for i in range(16):
            assert torch.equal(torch.stack(out["matched_indices"][0][i]), t(f[f"match_{i}"])), i
        for k in ("loss_ce", "loss_giou", "loss_counter"):
            assert maxerr(loss[k].reshape(()), f[f"loss.{k}"].reshape(())) <= 1e-3, k
        seq, ref_seq = out["seq"].cpu(), t(f["seq"].astype(np.int64))
        assert seq.shape == ref_seq.shape == (16, 300, int(f["max_caption_len"]))
        same = (seq == ref_seq)
        row_same = same.all(-1)
        # near-tie flips change the rest of a caption: at most 0.5 % of the 4800 captions, >= 99.9 % of the tokens
        assert float(same.float().mean()) >= 0.999 and int((~row_same).sum()) <= 24, (float(same.float().mean()), int((~row_same).sum()))
        lp = out["caption_probs"]["cap_prob_eval"].float().cpu()[:, ::4]
        keep = row_same[:, ::4]
        assert maxerr(lp[keep], t(f["cap_prob_eval"])[keep]) <= 2e-3


def test_headline_batch_of_16_eval_under_autocast_against_the_fp32_reference(monkeypatch):
    """The same batch under torch.autocast(bfloat16) against the reference's FP32 run: how far reduced-precision inference is from
    the reference at the benchmark's batch, for the default policy (one fp16 matrix-core product per fp32 product,
    gvl_amd/pdvc.py) -- eager and through GraphedEvalForward -- and, beside it, for the bf16-storage policy.  Measured (mean / max):
    boxes 3.3e-4 / 5.3e-3, logits 3.1e-3 / 4.4e-2, 91 % of the 144 000 greedy tokens equal -- bf16 storage: boxes 7.4e-3 / 1.3e-1,
    logits 7.2e-2 / 1.37, 45 % of the tokens; the mean errors must be at least 3x below the bf16-storage policy's."""
    from gvl_amd.config import make_opt
    from gvl_amd.parallel import GraphedEvalForward
    from gvl_amd.pdvc import build
    monkeypatch.delenv("GVL_AUTOCAST_INFERENCE", raising=False)
    dev = torch.device("cuda:0")
    f = load("pdvc_anet_full_b16")
    opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
    model, criterion, _, _ = build(opt)
    model.load_state_dict(pdvc_state(f, seed=100), strict=True)
    model = model.to(dev).eval()
    dt = to_dev(pdvc_dt(f, feat=int(f["feature_dim"]), seed=16), dev)

    def errors(out):
        e = {}
        for k, name in (("pred_boxes", "boxes"), ("pred_logits", "logits"), ("pred_count", "count")):
            d = (out[k].float().cpu() - t(f[k])).abs()
            e[name], e[name + "_mean"] = float(d.max()), float(d.mean())
        e["tokens"] = float((out["seq"].cpu() == t(f["seq"].astype(np.int64))).float().mean())
        return e
    graphed = GraphedEvalForward(model, criterion, autocast_dtype=torch.bfloat16)
    stats = {}
    for mode in ("eager", "graph"):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "eager"):
            out, loss = model(dt, criterion, None, "queries", eval_mode=True) if mode == "eager" else graphed(dt)
        assert out["pred_logits"].dtype == torch.float32
        e = stats[mode] = errors(out)
        assert e["boxes"] <= 2e-2 and e["logits"] <= 1.5e-1 and e["count"] <= 6e-2 and e["tokens"] >= 0.85, e
        for k in ("loss_ce", "loss_giou", "loss_counter"):             # (loss_ce is a sum over 4800 queries x the focal weight: ~60)
            assert maxerr(loss[k].reshape(()), f[f"loss.{k}"].reshape(())) <= 2e-2 * max(1.0, abs(float(f[f"loss.{k}"]))), k     # (a matched pair may differ on a cost near-tie)
    assert stats["eager"] == stats["graph"]                 # the captured step runs the same single-product kernels
    with torch.no_grad():
        exact, _ = model(dt, criterion, None, "queries", eval_mode=True)
    assert errors(exact)["logits"] < stats["eager"]["logits"]                # ... and they ARE the reduced product
    monkeypatch.setenv("GVL_AUTOCAST_INFERENCE", "bf16")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        lowp, _ = model(dt, criterion, None, "queries", eval_mode=True)
    bf = errors(lowp)
    print("autocast vs the fp32 reference at B = 16 -- single-product:", stats["eager"], "| bf16 storage:", bf)
    for k in ("boxes_mean", "logits_mean", "count_mean"):
        assert stats["eager"][k] * 3 <= bf[k], (k, stats["eager"][k], bf[k])


def test_graphed_eval_forward_follows_parameter_updates(built):
    """operands derived from weights are cached by parameter version and captured as graph constants: after an
    in-place parameter update the graphed forward must agree with the eager one again (new capture)."""
    from gvl_amd.parallel import GraphedEvalForward
    f, model, criterion, dev = built
    g = GraphedEvalForward(model, criterion)
    dt = to_dev(pdvc_dt(f), dev)
    g(dt)
    cap = model.caption_head[-1]
    saved = [p_.detach().clone() for p_ in (cap.core.h2att.weight, cap.embed.weight)]
    try:
        with torch.no_grad():
            cap.core.h2att.weight.mul_(1.3)
            cap.embed.weight.add_(0.05)
            ref_out, _ = model(dt, criterion, None, "queries", eval_mode=True)
        out, _ = g(dt)
        assert torch.equal(out["seq"], ref_out["seq"])
        assert maxerr(out["caption_probs"]["cap_prob_eval"], ref_out["caption_probs"]["cap_prob_eval"]) < 1e-5
        assert not torch.equal(out["seq"].cpu(), t(f["cuda.seq"]))          # the update really changed the captions
        assert len(g.graphs) == 1                                          # the graph of the old weights was dropped
    finally:
        with torch.no_grad():
            cap.core.h2att.weight.copy_(saved[0])
            cap.embed.weight.copy_(saved[1])


def test_position_embedding_kernel_equals_torch_formulation(built):
    """gvl_pos_embed_sine_f32 (one launch per pyramid level) against the PyTorch op sequence that mirrors
    position_encoding.py:38-64, incl. the gradient of the duration embedding; ragged masks, T = 1 and T > blockDim."""
    f, model, criterion, dev = built
    pe = model.base_encoder.pos_embed
    g = torch.Generator().manual_seed(2)
    for T in (1, 13, 100, 700):
        N = 3
        valid = torch.randint(1, T + 1, (N,), generator=g)
        mask = (torch.arange(T)[None] >= valid[:, None]).to(dev)
        mask[0] = False
        duration = torch.tensor([30.0, 200.0, 255.9], device=dev)
        x = torch.zeros(N, 512, T, device=dev)
        pe.zero_grad()
        out = pe(x, mask, duration)
        gout = torch.randn(out.shape, generator=g).to(dev)
        (out * gout).sum().backward()
        grads = [p_.grad.clone() for p_ in pe.parameters()]
        pe.zero_grad()
        with torch.autocast("cuda", dtype=torch.float32):      # autocast on -> the module takes its PyTorch formulation
            ref = pe(x, mask, duration)
        (ref * gout).sum().backward()
        assert out.shape == ref.shape == (N, 512, T)
        assert maxerr(out, ref) <= 2e-6
        for a_, p_ in zip(grads, pe.parameters()):
            assert maxerr(a_, p_.grad) <= 1e-4 * max(1.0, float(p_.grad.abs().max()))


@pytest.mark.parametrize("policy", ["bf16", "fp32", "f16"])
def test_graphed_eval_under_autocast_equals_eager_autocast(built_yc2, policy, monkeypatch):
    """the captured forward under torch.autocast(bfloat16) reproduces the eager autocast forward (same kernels, same
    casts) on the long-video batch -- for the bf16-storage path, the fp32 island and the default single-product policy"""
    from gvl_amd.parallel import GraphedEvalForward
    f, model, criterion, dev = built_yc2
    monkeypatch.setenv("GVL_AUTOCAST_INFERENCE", "" if policy == "f16" else policy)
    dt = to_dev(pdvc_dt(f, feat=int(f["feature_dim"]), seed=4), dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        ref_out, ref_loss = model(dt, criterion, None, "queries", eval_mode=True)
    g = GraphedEvalForward(model, criterion, autocast_dtype=torch.bfloat16)
    for _ in range(2):
        out, loss = g(dt)
        for k in ("pred_logits", "pred_boxes", "pred_count"):
            assert out[k].dtype == ref_out[k].dtype and maxerr(out[k].float(), ref_out[k].float()) < 1e-6, k
        assert torch.equal(out["seq"], ref_out["seq"])


def test_graphed_bf16_train_step_stays_finite():
    """Regression: the captured train step under torch.autocast(bfloat16) at the yc2 long-video shape produced NaN
    parameters after two replays when the capture ran on a different stream than the warm-up steps (autograd's
    AccumulateGrad nodes then formed a branch of the graph the optimizer kernels did not wait for).  Ten replays must
    stay finite and reduce the loss."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synth_batch
    from gvl_amd.config import make_opt
    from gvl_amd.parallel import GraphedTrainStep
    from gvl_amd.pdvc import build
    dev = torch.device("cuda:0")
    opt = make_opt("yc2_tsn_dvc", num_queries=100, frame_embedding_num=512, device="cuda")
    torch.manual_seed(0)
    model, criterion, _, _ = build(opt)
    model = model.to(dev).train()
    dt = synth_batch(16, 512, opt.feature_dim, opt.vocab_size, 3, dev)
    step = GraphedTrainStep(model, criterion, opt, warmup=1, autocast_dtype=torch.bfloat16)
    losses = []
    for _ in range(10):
        final, _ = step(dt)
        losses.append(final.clone())                  # the graph output is a static tensor
    losses = [float(x) for x in torch.stack(losses)]          # one read at the end: no synchronisation between replays
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0]
    assert all(torch.isfinite(p_).all() for p_ in model.parameters())


def test_graphed_bf16_train_step_survives_a_device_synchronisation():
    """Round-4 regression: in bench.py's bf16 train half (cfg A, caption-width buckets: 7 graphs over 8 rotating batches) the
    parameters went NaN two replays after the torch.cuda.synchronize() between warm-up and timed region -- every run, while
    the same sequence without the synchronisation, and the eager step, stayed finite.  Cause: ROCm 7.2's graph "packet
    capture" replay path mis-orders non-kernel nodes (hipMemsetAsync of PyTorch's reduction semaphores, D2D copies) against the
    kernel packets once the queue has gone idle; gvl_amd / bench.py set DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before the first HIP
    call.  This test replays that exact sequence and asserts finiteness, and that the switch is in the environment."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import rotating_batches
    from gvl_amd.config import make_opt
    from gvl_amd.parallel import GraphedTrainStep
    from gvl_amd.pdvc import build
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"          # set by `import gvl_amd` (conftest, before CUDA init)
    dev = torch.device("cuda:0")
    opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
    torch.manual_seed(0)
    model, criterion, _, _ = build(opt)
    model = model.to(dev).train()
    batches = rotating_batches(8, 16, 100, opt.feature_dim, opt.vocab_size, dev, seed=1)
    step = GraphedTrainStep(model, criterion, opt, autocast_dtype=torch.bfloat16, cap_len_policy="bucket")
    losses = []
    for i in list(range(8)) + [0, 1, 2]:
        losses.append(step(batches[i])[0].clone())
    torch.cuda.synchronize()
    torch.cuda.synchronize()
    for i in (0, 1, 2, 3, 4, 5):
        losses.append(step(batches[i])[0].clone())
    torch.cuda.synchronize()
    for i in (6, 7, 0):
        losses.append(step(batches[i])[0].clone())
    losses = [float(x) for x in torch.stack(losses)]
    assert all(np.isfinite(losses)), losses
    assert all(torch.isfinite(p_).all() for p_ in model.parameters())
