"""The inference layers of gvl_amd/layers.py (gvl_layers.hip: gvl_linear_f16x3_f32, gvl_layer_norm_rows_f32,
gvl_row_absmax_f32, gvl_msda1d_fused_forward_amax_f32) against fp64 PyTorch formulations of the reference's layer
arithmetic (pdvc/deformable_transformer.py:189-199,257-280; pdvc/ops/modules/ms_deform_attn.py:95-125).

Accuracy bar for the split-fp16 product: error against the fp64 product no larger than the fp32 library GEMM's own
(rms <= 1.05 x, max <= 1.5 x + tiny), as tests/test_gpu_gemm16.py demands of the captioner's products."""
import numpy as np
import pytest
import torch

from helpers import maxerr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _errs(got, want64):
    d = (got.double() - want64)
    return float(d.pow(2).mean().sqrt()), float(d.abs().max())


@pytest.mark.parametrize("R,K,N", [(4800, 512, 512), (3008, 512, 2048), (4800, 2048, 512), (300, 512, 256), (37, 512, 64),
                                   (129, 64, 128)])
def test_linear_matches_fp64_at_fp32_gemm_accuracy(R, K, N):
    from gvl_amd import layers as L
    x = _rand(R, K, seed=1) * torch.exp2(_rand(R, 1, seed=2, scale=3.0)).clamp(max=1e4)     # rows of very different size
    w, b = _rand(N, K, seed=3, scale=K ** -0.5), _rand(N, seed=4)
    am, _ = L.row_absmax(x)
    assert torch.equal(am, x.abs().amax(1))
    out = torch.empty(R, N, device=DEV)
    L.linear(x, L.Weights([(w, b)]), [L.seg(0, out, am)])
    want = x.double() @ w.double().t() + b.double()
    lib = torch.nn.functional.linear(x, w, b)
    rms, mx = _errs(out, want)
    rms32, mx32 = _errs(lib, want)
    assert rms <= 1.05 * rms32 + 1e-12 and mx <= 1.5 * mx32 + 1e-9, (rms, rms32, mx, mx32)


def test_linear_segments_addend_mask_relu_residual_and_row_maxima():
    """one launch, three column segments of a concatenated weight: [value_proj (masked rows) | projection of x + pos |
    ReLU + residual with row maxima] -- ms_deform_attn.py:95-100 and deformable_transformer.py:189-191 in one call"""
    from gvl_amd import layers as L
    R, K = 1507, 512
    x, pos = _rand(R, K, seed=5), _rand(R, K, seed=6, scale=2.0)
    w0, b0 = _rand(512, K, seed=7, scale=0.05), _rand(512, seed=8)
    w1, b1 = _rand(256, K, seed=9, scale=0.05), _rand(256, seed=10)
    w2, b2 = _rand(100, K, seed=11, scale=0.05), _rand(100, seed=12)           # padded to 128 columns
    res = _rand(R, 128, seed=13)
    mask = (torch.arange(R, device=DEV) % 7 == 3)
    am, amp = L.row_absmax(x, pos)
    assert torch.equal(amp, (x + pos).abs().amax(1))
    W = L.Weights([(w0, b0), (w1, b1), (w2, b2)])
    assert W.starts == [0, 512, 768] and W.N == 896
    o0, o1, o2 = (torch.full((R, n), float("nan"), device=DEV) for n in (512, 256, 128))
    am2 = torch.zeros(R, device=DEV)
    L.linear(x, W, [L.seg(0, o0, am, rowmask=mask.view(torch.uint8)), L.seg(512, o1, amp, addend=True),
                    L.seg(768, o2, am, resid=res, relu=True, amax_out=am2)], a2=pos)
    xd, pd = x.double(), pos.double()
    want0 = (xd @ w0.double().t() + b0.double()).masked_fill(mask[:, None], 0.0)
    want1 = (xd + pd) @ w1.double().t() + b1.double()
    want2 = res.double()[:, :100] + torch.relu(xd @ w2.double().t() + b2.double())
    assert maxerr(o0, want0) <= 2e-6 * float(want0.abs().max())
    assert torch.equal(o0[mask], torch.zeros_like(o0[mask]))
    assert maxerr(o1, want1) <= 2e-6 * float(want1.abs().max())
    assert maxerr(o2[:, :100], want2) <= 2e-6 * float(want2.abs().max())
    assert torch.equal(o2[:, 100:], res[:, 100:])                             # padded weight rows: relu(0) + residual
    assert torch.equal(am2, o2.abs().amax(1))                                 # the epilogue's atomic row maxima are exact
    # the addend may be shared by blocks of rows (query_pos: the same embedding for every video)
    q = _rand(137, K, seed=14)
    _, ampq = L.row_absmax(x, q)
    o3 = torch.empty(R, 256, device=DEV)
    L.linear(x, L.Weights([(w1, b1)]), [L.seg(0, o3, ampq, addend=True)], a2=q)
    want3 = (xd + q.double()[torch.arange(R, device=DEV) % 137]) @ w1.double().t() + b1.double()
    assert maxerr(o3, want3) <= 2e-6 * float(want3.abs().max())


def test_linear_non_finite_and_wide_range_rows():
    """a non-finite element makes ITS output row non-finite and no other; elements 2^-40 below the
    row maximum are below the representable floor (documented domain) and cost at most K 2^-33 amax max|w|"""
    from gvl_amd import layers as L
    R, K, N = 256, 512, 128
    x, w = _rand(R, K, seed=20), _rand(N, K, seed=21, scale=0.05)
    x[5, 17] = float("inf")
    x[9, 300] = float("nan")
    x[11] = x[11] * 2.0 ** -40
    x[11, 3] = 1.0                                                           # one large element, the rest 2^-40 below it
    am, _ = L.row_absmax(x)
    out = torch.empty(R, N, device=DEV)
    L.linear(x, L.Weights([(w, None)]), [L.seg(0, out, am)])
    fin = torch.isfinite(out).all(1)
    assert not fin[5] and not fin[9] and int((~fin).sum()) == 2
    want = x.double() @ w.double().t()
    ok = torch.ones(R, dtype=torch.bool, device=DEV)
    ok[5] = ok[9] = False
    bound = 2.0 ** -21 * (x.double().abs() @ w.double().abs().t()) + K * 2.0 ** -33 * am.double()[:, None] * float(w.abs().max())
    assert bool(((out.double() - want).abs()[ok] <= bound[ok] + 1e-30).all())


@pytest.mark.parametrize("R,C", [(4800, 512), (3008, 512), (33, 256), (7, 1024)])
def test_layer_norm_rows(R, C):
    from gvl_amd import layers as L
    x, pos = _rand(R, C, seed=30, scale=3.0) + 0.7, _rand(300 if R > 300 else R, C, seed=31)
    norm = torch.nn.LayerNorm(C).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(_rand(C, seed=32) * 0.3 + 1.0)
        norm.bias.copy_(_rand(C, seed=33) * 0.2)
    y, am, amp = L.layer_norm(x, norm, pos=pos)
    want = torch.nn.functional.layer_norm(x.double(), (C,), norm.weight.double(), norm.bias.double(), norm.eps)
    lib = norm(x)
    assert maxerr(y, want) <= 2.0 * maxerr(lib, want) + 1e-6
    assert torch.equal(am, y.abs().amax(1))
    assert torch.equal(amp, (y + pos[torch.arange(R, device=DEV) % pos.shape[0]]).abs().amax(1))


def test_sampling_kernel_leaves_exact_row_maxima():
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    from helpers import level_lengths
    B, M, D, L_, P, Q, T = 4, 8, 64, 4, 4, 300, 100
    lens = level_lengths(T)
    S = sum(lens)
    value = _rand(B, S, M, D, seed=40)
    proj = _rand(B, Q, 2 * M * L_ * P, seed=41, scale=0.7)
    ref = torch.rand(B, Q, L_, 1, generator=torch.Generator().manual_seed(42)).to(DEV)
    shapes = torch.tensor([(1, x) for x in lens], dtype=torch.long, device=DEV)
    lsi = torch.tensor(np.concatenate([[0], np.cumsum(lens)[:-1]]), dtype=torch.long, device=DEV)
    MSDA.attach_host_shapes(shapes, lsi, [(1, x) for x in lens], [int(v) for v in lsi.tolist()])
    plain = MSDA.msda1d_fused_forward(value, shapes, lsi, proj, ref, L_, P)
    am = torch.zeros(B * Q, device=DEV)
    out = MSDA.msda1d_fused_forward(value, shapes, lsi, proj, ref, L_, P, amax_out=am)
    assert torch.equal(out, plain)
    assert torch.equal(am, out.abs().amax(-1).reshape(-1))


@pytest.mark.parametrize("B,Q,T", [(16, 300, 100), (3, 37, 64), (2, 300, 512)])
def test_sampling_kernel_with_one_set_of_operand_rows_for_every_video(B, Q, T):
    """gvl_msda1d_fused_forward_shared_amax_f32: proj (1, Q, .) read by every video == the launch on the rows copied B times"""
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    from helpers import level_lengths
    M, D, L_, P = 8, 64, 4, 4
    lens = level_lengths(T)
    S = sum(lens)
    value = _rand(B, S, M, D, seed=43)
    proj_q = _rand(1, Q, 2 * M * L_ * P, seed=44, scale=0.7)
    ref = torch.rand(B, Q, L_, 1, generator=torch.Generator().manual_seed(45)).to(DEV)
    shapes = torch.tensor([(1, x) for x in lens], dtype=torch.long, device=DEV)
    lsi = torch.tensor(np.concatenate([[0], np.cumsum(lens)[:-1]]), dtype=torch.long, device=DEV)
    MSDA.attach_host_shapes(shapes, lsi, [(1, x) for x in lens], [int(v) for v in lsi.tolist()])
    am_w, am_g = torch.zeros(B * Q, device=DEV), torch.zeros(B * Q, device=DEV)
    want = MSDA.msda1d_fused_forward(value, shapes, lsi, proj_q.expand(B, -1, -1).contiguous(), ref, L_, P, amax_out=am_w)
    got = MSDA.msda1d_fused_forward(value, shapes, lsi, proj_q, ref, L_, P, amax_out=am_g)
    assert torch.equal(got, want) and torch.equal(am_g, am_w)


@pytest.mark.parametrize("B,Q,H,masked", [(16, 300, 8, False), (3, 300, 8, True), (2, 37, 4, True), (1, 320, 2, False)])
def test_attention_core_matches_fp64(B, Q, H, masked):
    """gvl_mha_core_f32 vs softmax(q k^T / sqrt(64)) v in fp64 (nn.MultiheadAttention's core, deformable_transformer.py:266-270)"""
    from gvl_amd import layers as L
    C = 64 * H
    qkv = _rand(B * Q, 3 * C, seed=60, scale=1.5)
    keep = None
    if masked:
        keep = torch.ones(B, Q, dtype=torch.bool, device=DEV)
        for b in range(B):
            keep[b, Q - 1 - 7 * b:] = False
            keep[b, 3 + b] = False
    am = torch.zeros(B * Q, device=DEV)
    out = L.mha_core(qkv, B, Q, H, keep, am)
    t = qkv.double().view(B, Q, 3, H, 64).permute(2, 0, 3, 1, 4)
    sc = t[0] @ t[1].transpose(-1, -2) / 8.0
    if keep is not None:
        sc = sc.masked_fill(~keep[:, None, None, :], float("-inf"))
    want = (torch.softmax(sc, -1) @ t[2]).transpose(1, 2).reshape(B * Q, C)
    lib = torch.nn.functional.scaled_dot_product_attention(
        *(x.float() for x in t), attn_mask=keep[:, None, None, :] if keep is not None else None).transpose(1, 2).reshape(B * Q, C)
    assert maxerr(out, want) <= 2.0 * maxerr(lib, want) + 2e-6
    assert torch.equal(am, out.abs().amax(1))


def test_encoder_geometry_equals_the_pytorch_formulation():
    """valid ratios and encoder reference points from the flat mask in one launch == deformable_transformer.py:81-83,209-218"""
    from gvl_amd import layers as L
    from gvl_amd.deformable_transformer import DeformableTransformerEncoder, make_level_tensors
    from helpers import level_lengths
    B, T = 5, 100
    lens = level_lengths(T)
    starts = [int(v) for v in np.concatenate([[0], np.cumsum(lens)[:-1]])]
    masks = []
    for l_, n in enumerate(lens):
        m = torch.zeros(B, n, dtype=torch.bool, device=DEV)
        for b in range(B):
            m[b, max(1, n - (b * n) // 6):] = True
        masks.append(m)
    vr, ref = L.encoder_geometry(torch.cat(masks, 1), lens, starts)
    want_vr = torch.stack([torch.sum(~m, 1).float() / m.shape[1] for m in masks], 1)
    assert torch.equal(vr, want_vr)
    ts, _ = make_level_tensors(lens, torch.device(DEV))
    want_ref = DeformableTransformerEncoder.get_reference_points(ts, want_vr, torch.device(DEV))
    assert ref.shape == want_ref.shape and maxerr(ref, want_ref) <= 1e-7


@pytest.mark.parametrize("N,T,Cin", [(16, 100, 512), (3, 61, 96), (2, 512, 96)])
def test_flat_base_encoder_equals_the_pytorch_formulation(N, T, Cin, monkeypatch):
    """BaseEncoder.forward_flat (conv1d levels as strided-view products, GroupNorm into the flattened layout, one geometry
    launch) == BaseEncoder.forward + DeformableTransformer.prepare_encoder_inputs (base_encoder.py:55-82,
    position_encoding.py:38-64, deformable_transformer.py:85-115)"""
    from gvl_amd.base_encoder import BaseEncoder
    torch.manual_seed(5)
    enc = BaseEncoder(4, Cin, 512).to(DEV).eval()
    with torch.no_grad():
        for seq in enc.input_proj:
            seq[0].bias.normal_(0, 0.1)
            seq[1].weight.normal_(1.0, 0.2)
            seq[1].bias.normal_(0, 0.2)
    tr = _transformer()
    vf = _rand(N, T, Cin, seed=70)
    mask = torch.zeros(N, T, dtype=torch.bool, device=DEV)
    for b in range(N):
        mask[b, T - (b * T) // (2 * N):] = True
    dur = torch.tensor([30.0 + 13.7 * b for b in range(N)], device=DEV)
    with torch.no_grad():
        assert enc.flat_eligible(vf, mask)
        src, mflat, pos, lengths = enc.forward_flat(vf, mask, dur, tr.level_embed)
        tsh, lsi, vr = tr.flat_geometry(mflat, lengths)
        srcs, masks, poses = enc(vf, mask, dur)
        monkeypatch.setenv("GVL_LAYERS", "torch")
        w_src, w_tsh, w_lsi, w_vr, w_pos, w_mask = tr.prepare_encoder_inputs(srcs, masks, poses)
    assert lengths == [int(x) for x in w_tsh.tolist()] and torch.equal(lsi, w_lsi)
    assert torch.equal(mflat, w_mask) and torch.equal(vr, w_vr)
    assert maxerr(src, w_src) <= 2e-5 * float(w_src.abs().max())
    assert maxerr(pos, w_pos) <= 1e-6 * max(1.0, float(w_pos.abs().max()))


def _transformer(seed=0):
    from gvl_amd.deformable_transformer import DeformableTransformer
    torch.manual_seed(seed)
    tr = DeformableTransformer(d_model=512, nhead=8, num_encoder_layers=2, num_decoder_layers=2, dim_feedforward=2048,
                               dropout=0.1, return_intermediate_dec=True, num_feature_levels=4)
    with torch.no_grad():                      # the reference initialises offsets / logits to constants: make them live
        for m in tr.modules():
            if hasattr(m, "sampling_offsets"):
                m.sampling_offsets.weight.normal_(0, 0.02)
                m.attention_weights.weight.normal_(0, 0.05)
    return tr.to(DEV).eval()


@pytest.mark.parametrize("B,T,masked", [(16, 100, False), (3, 60, True)])
def test_fused_inference_layers_equal_the_pytorch_formulation(B, T, masked, monkeypatch):
    """encoder + decoder (with a box MLP for the iterative refinement) through gvl_amd/layers.py vs the same modules
    through their PyTorch formulation (GVL_LAYERS=torch), and vs an fp64 evaluation of that formulation"""
    from gvl_amd.deformable_transformer import make_level_tensors
    from gvl_amd.pdvc import MLP
    from helpers import level_lengths
    tr = _transformer()
    C, Q = 512, 300
    lens = level_lengths(T)
    S = sum(lens)
    src, pos = _rand(B, S, C, seed=50), _rand(B, S, C, seed=51)
    tshapes, lsi = make_level_tensors(lens, torch.device(DEV))
    mask = torch.zeros(B, S, dtype=torch.bool, device=DEV)
    vr = torch.ones(B, 4, device=DEV)
    if masked:
        for b in range(B):
            for l_, (st, n) in enumerate(zip(np.concatenate([[0], np.cumsum(lens)[:-1]]), lens)):
                keep = max(1, int(n * (0.5 + 0.2 * b)))
                mask[b, st + keep:st + n] = True
                vr[b, l_] = keep / n
    torch.manual_seed(3)
    box = torch.nn.ModuleList([MLP(C, C, 2, 3), MLP(C, C, 2, 3)]).to(DEV).eval()
    tr.decoder.bbox_head = box
    qe = _rand(Q, 2 * C, seed=52)
    qmask = torch.ones(B, Q, dtype=torch.bool, device=DEV)

    def run():
        with torch.no_grad():
            memory = tr.forward_encoder(src, tshapes, lsi, vr, pos, mask)
            ref0, tgt, ref, qpos = tr.prepare_decoder_input_query(memory, qe)
            hs, refs = tr.forward_decoder(tgt, ref, memory, tshapes, lsi, vr, qpos, mask, qmask, False)
        return memory, hs, refs

    from gvl_amd import layers as L
    got = run()
    assert tr.decoder.__dict__.get("_gvl_deltas") is not None and L.enabled()
    monkeypatch.setenv("GVL_LAYERS", "torch")
    want = run()
    assert tr.decoder.__dict__.get("_gvl_deltas") is None
    for name, a, b in zip(("memory", "hs", "refs"), got, want):
        tol = 2e-4 if name != "refs" else 2e-5
        assert maxerr(a, b) <= tol * max(1.0, float(b.abs().max())), (name, maxerr(a, b))


def test_query_constants_kept_for_inference_follow_their_parameters():
    """'queries' input (deformable_transformer.py:128-135): the decoder's first residual rows, the positional rows, their row
    maxima and sigmoid(reference_points(query_pos)) depend on parameters only and are kept across inference forwards
    (gvl_amd.layers._query_rows, DeformableTransformer.prepare_decoder_input_query) -- until a parameter is written."""
    from gvl_amd import layers as L
    from gvl_amd.deformable_transformer import DeformableTransformer
    B, Q, C = 3, 70, 512
    emb = _rand(Q, 2 * C, seed=5)
    lin = torch.nn.Linear(C, 1).to(DEV)

    class Shell(torch.nn.Module):                    # the method only touches self.reference_points / self.__dict__
        prepare_decoder_input_query = DeformableTransformer.prepare_decoder_input_query
    sh = Shell().to(DEV)
    sh.reference_points = lin
    mem = torch.empty(B, 1, C, device=DEV)
    with torch.no_grad():
        ref1, tgt, _, qpos = sh.prepare_decoder_input_query(mem, emb)
        ref2 = sh.prepare_decoder_input_query(mem, emb)[0]
        assert ref2 is ref1                                            # kept
        want = torch.sigmoid(torch.nn.functional.linear(emb[:, :C], lin.weight, lin.bias))[None].expand(B, -1, -1)
        assert maxerr(ref1, want) < 1e-6
        lin.bias.add_(0.5)                                             # a parameter is written: recomputed
        ref3 = sh.prepare_decoder_input_query(mem, emb)[0]
        assert ref3 is not ref1 and maxerr(ref3, torch.sigmoid(torch.logit(want) + 0.5)) < 1e-5
        dec = torch.nn.Module()
        rows = L._query_rows(dec, tgt, qpos, B, Q, C)
        assert rows is not None and L._query_rows(dec, tgt, qpos, B, Q, C) is rows
        x, qp, am_x, am_xp = rows
        assert torch.equal(x.view(B, Q, C), tgt) and torch.equal(qp, qpos[0])
        assert torch.equal(am_x, x.abs().amax(1)) and torch.equal(am_xp.view(B, Q), (tgt + qpos).abs().amax(2))
        emb.mul_(2.0)                                                  # the embedding is written: new rows
        rows2 = L._query_rows(dec, tgt, qpos, B, Q, C)
        assert rows2 is not rows and torch.equal(rows2[0].view(B, Q, C), tgt)
    with torch.enable_grad():                                          # training keeps the reference's own evaluation
        assert L._query_rows(dec, tgt, qpos, B, Q, C) is None


def test_first_decoder_layer_constants_are_kept_and_equal_the_recomputed_forward(monkeypatch):
    """'queries' input, no padded query: the first decoder layer's self-attention block and offsets / weights projection depend on
    parameters only and are kept across inference forwards (gvl_amd.layers._first_layer_constants) -- bit for bit the forward
    that recomputes them (GVL_FIRST_LAYER_CACHE=0), for a second batch too, and recomputed after a parameter changes"""
    from helpers import load, pdvc_dt, pdvc_state
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    f = load("pdvc_anet_full")
    opt = make_opt("anet_tsp_ssvg", num_queries=300, frame_embedding_num=100, device="cuda")
    model, criterion, _, _ = build(opt)
    model.load_state_dict(pdvc_state(f, seed=100), strict=True)
    model = model.to(DEV).eval()
    from bench import synth_batch
    dts = [synth_batch(4, 100, opt.feature_dim, opt.vocab_size, [2, 0, 3, 1], DEV, seed=70 + i) for i in range(2)]
    dec = model.transformer.decoder

    def run(dt):
        with torch.no_grad():
            out, _ = model(dt, None, None, "queries", eval_mode=True)
        return out["pred_logits"].clone(), out["pred_boxes"].clone(), out["seq"].clone()
    monkeypatch.setenv("GVL_FIRST_LAYER_CACHE", "0")
    want = [run(dt) for dt in dts]
    monkeypatch.setenv("GVL_FIRST_LAYER_CACHE", "1")
    dec.__dict__.pop("_gvl_first_layer", None)
    got = [run(dt) for dt in dts] + [run(dts[0])]
    assert len(dec.__dict__["_gvl_first_layer"]) == 1
    for g_, w_ in zip(got, want + [want[0]]):
        assert all(torch.equal(a, b) for a, b in zip(g_, w_))
    with torch.no_grad():
        dec.layers[0].norm2.bias.add_(0.01)                        # a parameter of the kept block changes: recomputed
    kept = dict(dec.__dict__["_gvl_first_layer"])
    moved = run(dts[0])
    assert not torch.equal(moved[0], want[0][0])
    assert len(dec.__dict__["_gvl_first_layer"]) == 1 and set(dec.__dict__["_gvl_first_layer"]) != set(kept)


def test_split_k_product_with_bias_over_overlapping_rows_matches_float64():
    """gvl_linear_f16x3_splitk_bias_f32: the k = 3, stride 2 convolution of the feature pyramid as a product over rows of taps that
    OVERLAP (row stride 2 C_in < 3 C_in) with the bias added to the finished sum -- against float64, and the plain kernel"""
    from gvl_amd import layers as L
    torch.manual_seed(2)
    N, T1, ch, C = 16, 51, 512, 512
    xp = torch.randn(N + 1, 2 * T1, ch, device=DEV)
    a = xp.as_strided((N * T1, 3 * ch), (2 * ch, 1))
    w = torch.randn(C, 3 * ch, device=DEV) * 0.03
    b = torch.randn(C, device=DEV)
    W = L.Weights([(w, b)])
    assert L.splitk_pays(a.shape[0], W.N, W.K)
    am, _ = L.row_absmax(a)
    y = L.linear_splitk(a, am, W, bias=True)
    y0 = torch.empty_like(y)
    L.linear(a, W, [L.seg(0, y0, am)])
    ref = a.double() @ w.double().t() + b.double()
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= 2e-6 * scale and float((y0.double() - ref).abs().max()) <= 2e-6 * scale
    assert float((L.linear_splitk(a, am, W) .double() - (ref - b.double())).abs().max()) <= 2e-6 * scale       # without the bias
