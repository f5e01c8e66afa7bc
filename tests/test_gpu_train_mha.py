"""gvl_amd/train_mha.py (gvl_mha_train.hip): the decoder layer's nn.MultiheadAttention in training -- in-projection with the
positional addend, attention core with key mask and dropout on the weights, out-projection -- against float64 evaluations of
torch's own formulation.  p = 0: nn.MultiheadAttention itself is the reference.  p > 0: the mask the kernels draw is a documented
hash of (seed, step, b, h, q, k) (include/gvl_msda.h); the test recomputes it and holds output and gradients to the float64
attention UNDER THAT MASK."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _hash32(x):
    m = 0xFFFFFFFF
    x = x & m
    x = x ^ (x >> 16); x = (x * 0x7feb352d) & m
    x = x ^ (x >> 15); x = (x * 0x846ca68b) & m
    return x ^ (x >> 16)


def _keep_mask(B, H, Q, p, seed, step):
    key = _hash32(torch.tensor((seed + step * 0x9E3779B9) & 0xFFFFFFFF, dtype=torch.int64, device=DEV))
    idx = torch.arange(B * H * Q * Q, dtype=torch.int64, device=DEV)
    # (the kernel receives p as a C float: 0.1f 2^32 is 7 above 0.1 2^32 -- one mask element in 6e8 would differ, which fails
    #  this test at about one (seed, step) pair in fifty)
    thr = min(int(float(torch.tensor(p, dtype=torch.float32)) * 4294967296.0), 0xFFFFFFFF)
    return (_hash32(idx ^ key) >= thr).view(B, H, Q, Q)


def _ref_attention(x, pos, w, b, wo, bo, H, key_keep, drop_keep, p):
    """float64: (B, Q, C) -> (B, Q, C) as nn.MultiheadAttention(q = k = x + pos, v = x) computes it, dropout mask given"""
    B, Q, C = x.shape
    xq = x + pos
    q, k, v = xq @ w[:C].t() + b[:C], xq @ w[C:2 * C].t() + b[C:2 * C], x @ w[2 * C:].t() + b[2 * C:]
    sp = lambda t: t.view(B, Q, H, C // H).transpose(1, 2)                      # noqa: E731
    s = sp(q) @ sp(k).transpose(-1, -2) / (C // H) ** 0.5
    s = s.masked_fill(~key_keep[:, None, None, :], float("-inf"))
    a = torch.softmax(s, -1)
    if drop_keep is not None:
        a = a * drop_keep / (1.0 - p)
    o = (a @ sp(v)).transpose(1, 2).reshape(B, Q, C)
    return o @ wo.t() + bo


@pytest.mark.parametrize("B,Q,p,lead", [(16, 300, 0.0, 0), (16, 300, 0.1, 0), (2, 300, 0.25, 0), (4, 170, 0.1, 0), (16, 64, 0.0, 0),
                                        (4, 300, 0.0, 40), (4, 170, 0.1, 32)])
def test_self_attention_matches_float64(B, Q, p, lead):
    """lead > 0: the first `lead` keys of some videos are padded -- a whole leading 32-key tile without a valid key, which the
    online softmax must survive (torch's MHA returns finite values there)"""
    from gvl_amd import train_layers as TL
    from gvl_amd import train_mha as TM
    C, H = 512, 8
    torch.manual_seed(B * Q)
    mha = torch.nn.MultiheadAttention(C, H, dropout=p).to(DEV).train()
    with torch.no_grad():
        mha.in_proj_bias.normal_(0, 0.3)
        mha.out_proj.bias.normal_(0, 0.3)
    tgt = (torch.randn(B, Q, C, device=DEV) * 1.5).requires_grad_()
    emb = torch.randn(Q, 2 * C, device=DEV).requires_grad_()
    pos = emb[:, :C].unsqueeze(0).expand(B, -1, -1)
    mask = torch.ones(B, Q, dtype=torch.bool, device=DEV)
    for i in range(B):
        mask[i, Q - 1 - 7 * i % Q // 3:] = i % 3 == 0                              # ragged key masks, some videos unmasked
    mask[:, 0] = True
    if lead:
        mask[::2, :lead] = False                                                   # non-prefix key masks
    assert TM.eligible(mha, tgt, pos)
    TL.advance(DEV)
    step = int(TL.step_counter(DEV).item())
    out = TM.self_attention(mha, tgt, pos, mask)
    g = torch.randn(B, Q, C, device=DEV)
    out.backward(g)
    got = [out.detach(), tgt.grad, emb.grad, mha.in_proj_weight.grad, mha.in_proj_bias.grad, mha.out_proj.weight.grad,
           mha.out_proj.bias.grad]
    keep = None
    if p > 0:
        seed = TL._site_seed(mha.__dict__["_gvl_site_drop"])
        keep = _keep_mask(B, H, Q, p, seed, step).double()
        assert abs(float(keep.mean()) - (1 - p)) < 5e-3
    d = lambda t: t.detach().double().requires_grad_()                           # noqa: E731
    t64, e64 = d(tgt), d(emb)
    ps = [d(t) for t in (mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias)]
    ref = _ref_attention(t64, e64[:, :C].unsqueeze(0).expand(B, -1, -1), ps[0], ps[1], ps[2], ps[3], H, mask, keep, p)
    ref.backward(g.double())
    refs = [ref.detach(), t64.grad, e64.grad] + [t.grad for t in ps]
    names = ["out", "d tgt", "d query_embed", "d in_proj_weight", "d in_proj_bias", "d out_proj.weight", "d out_proj.bias"]
    for n, a, r in zip(names, got, refs):
        err = float((a.double() - r).abs().max() / r.abs().max())
        assert err < 2e-5, (n, err)
    if p == 0:
        # ... and nn.MultiheadAttention itself (fp32) agrees with the same float64 result no better
        o2 = mha(pos.detach().transpose(0, 1) + tgt.detach().transpose(0, 1), pos.detach().transpose(0, 1) + tgt.detach().transpose(0, 1),
                 tgt.detach().transpose(0, 1), key_padding_mask=~mask, need_weights=False)[0].transpose(0, 1)
        e_torch = float((o2.double() - refs[0]).abs().max() / refs[0].abs().max())
        e_own = float((got[0].double() - refs[0]).abs().max() / refs[0].abs().max())
        assert e_own < max(4 * e_torch, 2e-6), (e_own, e_torch)


def test_attention_dropout_follows_the_step_counter_and_its_own_forward():
    from gvl_amd import train_layers as TL
    from gvl_amd import train_mha as TM
    B, Q, C, H = 4, 300, 512, 8
    torch.manual_seed(0)
    mha = torch.nn.MultiheadAttention(C, H, dropout=0.2).to(DEV).train()
    tgt = torch.randn(B, Q, C, device=DEV, requires_grad=True)
    pos = torch.randn(Q, C, device=DEV).unsqueeze(0).expand(B, -1, -1)
    mask = torch.ones(B, Q, dtype=torch.bool, device=DEV)
    TL.advance(DEV)
    a = TM.self_attention(mha, tgt, pos, mask)
    a2 = TM.self_attention(mha, tgt, pos, mask)
    assert torch.equal(a, a2)                                    # same step: same masks (deterministic kernels)
    TL.advance(DEV)
    b = TM.self_attention(mha, tgt, pos, mask)
    assert not torch.equal(a, b)                                 # next step: other masks
    ga, = torch.autograd.grad(a.sum(), tgt, retain_graph=True)   # backward of the FIRST forward after the counter moved ...
    TL.advance(DEV)
    ga2, = torch.autograd.grad(a.sum(), tgt)
    assert torch.equal(ga, ga2)                                  # ... still regenerates the first forward's masks
    mha.eval()
    e1, e2 = TM.self_attention(mha, tgt, pos, mask), TM.self_attention(mha, tgt, pos, mask)
    assert torch.equal(e1, e2)


def test_decoder_layer_routes_training_attention_to_the_hand_written_kernels():
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    from gvl_amd.deformable_transformer import DeformableTransformerDecoderLayer, make_level_tensors
    torch.manual_seed(0)
    B, Q, C = 16, 300, 512
    layer = DeformableTransformerDecoderLayer(C, 512, 0.1, "relu", 4, 8, 4).to(DEV).train()
    lengths = [100, 50, 25, 13]
    ts, ls = make_level_tensors(lengths, DEV)
    S = sum(lengths)
    tgt = torch.randn(B, Q, C, device=DEV, requires_grad=True)
    qp = torch.randn(Q, C, device=DEV).unsqueeze(0).expand(B, -1, -1)
    ref = torch.rand(B, Q, 4, 2, device=DEV)
    src = torch.randn(B, S, C, device=DEV)
    MSDA.profile_enable(2)
    try:
        MSDA.profile_collect()
        out = layer(tgt, qp, ref, src, ts, ls, torch.zeros(B, S, dtype=torch.bool, device=DEV), torch.ones(B, Q, dtype=torch.bool, device=DEV))
        out.sum().backward()
        torch.cuda.synchronize()
        names = [t[0] for t in MSDA.profile_collect()]
    finally:
        MSDA.profile_enable(0)
    assert names.count("mha_train") == 3, names                  # forward, dk / dv, dq
    assert torch.isfinite(tgt.grad).all()


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_first_layer_in_projection_on_the_embedding_rows_equals_the_batch_expanded_one(p, monkeypatch):
    """the first decoder layer under the 'queries' input: q, k, v from the Q embedding rows, copied to the B videos, dqkv summed
    over the videos before the backward products (_InProjShared) -- against the same call on the batch-expanded rows
    (GVL_INPROJ_SHARED=0): output and every gradient, with and without attention dropout (same masks: same step)"""
    from gvl_amd import layers as LY
    from gvl_amd import train_layers as TL
    from gvl_amd import train_mha as TM
    torch.manual_seed(9)
    B, Q, C, H = 16, 300, 512, 8
    mha = torch.nn.MultiheadAttention(C, H, dropout=p).to(DEV).train()
    mask = torch.ones(B, Q, dtype=torch.bool, device=DEV)
    g = torch.randn(B, Q, C, device=DEV)
    res = []
    for shared in ("1", "0"):
        monkeypatch.setenv("GVL_INPROJ_SHARED", shared)
        emb = torch.randn(Q, 2 * C, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4)).requires_grad_()
        pos, tgt, pos_rows, tgt_rows = LY.expand_parts(emb, B, 2, rows=True)
        pos._gvl_rows, tgt._gvl_rows = pos_rows, tgt_rows
        mha.zero_grad(set_to_none=True)
        assert TM.eligible(mha, tgt, pos)
        TL.step_counter(DEV).fill_(7)
        TL.arena(DEV).snap = None
        out = TM.self_attention(mha, tgt, pos, mask)
        out.backward(g)
        res.append([out.detach().clone(), emb.grad.clone(), mha.in_proj_weight.grad.clone(), mha.in_proj_bias.grad.clone(),
                    mha.out_proj.weight.grad.clone()])
    for name, a, b in zip(("out", "d query_embed", "d in_proj_weight", "d in_proj_bias", "d out_proj.weight"), *res):
        err = float((a - b).abs().max() / b.abs().max())
        assert err < (1e-6 if name == "out" else 2e-5), (name, err)
