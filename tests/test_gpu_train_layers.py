"""gvl_amd/train_layers.py (gvl_train_layers.hip): norm(x + dropout(sub)) of the encoder / decoder layers in training
(pdvc/deformable_transformer.py:189-199, 266-280) as one forward and one backward kernel, against the PyTorch formulation in
fp64.  p = 0: exact LayerNorm(x + sub) and its gradients.  p > 0: the mask the kernel drew is recovered from its own output
(z = x + keep sub / (1 - p)), must have the right keep rate, differ between sites and steps, and the gradients must be those of
the PyTorch formulation UNDER THAT MASK."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _ref(x, sub, keep, p, w, b, eps):
    z = x.double() + keep.double() * sub.double() / (1.0 - p)
    return torch.nn.functional.layer_norm(z, (x.shape[-1],), w.double(), b.double(), eps)


@pytest.mark.parametrize("B,Q,C", [(16, 300, 512), (16, 188, 512), (3, 7, 64), (2, 129, 1024), (1, 1, 4)])
@pytest.mark.parametrize("layout", ["contiguous", "transposed_sub", "expanded_x"])
def test_no_dropout_is_layer_norm_of_the_sum(B, Q, C, layout):
    from gvl_amd import train_layers as TL
    norm = torch.nn.LayerNorm(C).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(_rand(C, seed=1) * 0.5 + 1.0)
        norm.bias.copy_(_rand(C, seed=2))
    drop = torch.nn.Dropout(0.1).eval()                      # eval: p = 0
    x0, s0 = _rand(B, Q, C, seed=3), _rand(B, Q, C, seed=4, scale=2.0)
    if layout == "expanded_x":
        x0 = _rand(Q, C, seed=3)
    x0.requires_grad_(True)
    s0.requires_grad_(True)
    x = x0.unsqueeze(0).expand(B, -1, -1) if layout == "expanded_x" else x0
    sub = s0.transpose(0, 1).contiguous().transpose(0, 1) if layout == "transposed_sub" else s0
    assert TL.eligible(x, sub, norm, drop)
    y = TL.residual_dropout_norm(x, sub, drop, norm)
    gy = _rand(B, Q, C, seed=5)
    y.backward(gy)
    got = [t_.grad.clone() for t_ in (x0, s0, norm.weight, norm.bias)]
    for t_ in (x0, s0, norm.weight, norm.bias):
        t_.grad = None
    x64, s64 = x0.detach().double().requires_grad_(True), s0.detach().double().requires_grad_(True)
    w64, b64 = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
    xx = x64.unsqueeze(0).expand(B, -1, -1) if layout == "expanded_x" else x64
    ref = torch.nn.functional.layer_norm(xx + s64, (C,), w64, b64, norm.eps)
    ref.backward(gy.double())
    assert float((y.detach().double() - ref.detach()).abs().max()) <= 5e-6
    for g_, r_ in zip(got, (x64.grad, s64.grad, w64.grad, b64.grad)):
        assert float((g_.double() - r_).abs().max()) <= 2e-5 * max(1.0, float(r_.abs().max()))


def test_dropout_mask_rate_sites_steps_and_gradients():
    from gvl_amd import train_layers as TL
    B, Q, C, p = 16, 300, 512, 0.1
    norm = torch.nn.LayerNorm(C).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(_rand(C, seed=1) * 0.5 + 1.0)
        norm.bias.copy_(_rand(C, seed=2))
    drop_a, drop_b = torch.nn.Dropout(p).train(), torch.nn.Dropout(p).train()
    x = _rand(B, Q, C, seed=3).requires_grad_(True)
    sub = (_rand(B, Q, C, seed=4).abs() + 0.5).requires_grad_(True)          # no zeros: the mask is readable from z

    def run(drop):
        """-> (y, keep mask recovered through the autograd graph's saved z)"""
        y = TL.residual_dropout_norm(x, sub, drop, norm)
        z = y.grad_fn.saved_tensors[0]
        return y, ((z - x.detach()).abs() > 1e-6)
    torch.manual_seed(11)
    TL.advance(DEV)
    y1, k1 = run(drop_a)
    rate = float(k1.float().mean())
    assert abs(rate - (1 - p)) < 2e-3                                        # 2.4 M draws: sigma = 2e-4
    assert abs(float(k1.float().mean(dim=(0, 1)).min()) - (1 - p)) < 0.03    # ... per column
    assert abs(float(k1.float().mean(dim=2).min()) - (1 - p)) < 0.08         # ... per row
    # forward and backward under THAT mask
    gy = _rand(B, Q, C, seed=5)
    y1.backward(gy)
    got = [t_.grad.clone() for t_ in (x, sub, norm.weight, norm.bias)]
    x64, s64 = x.detach().double().requires_grad_(True), sub.detach().double().requires_grad_(True)
    w64, b64 = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(x64 + k1.double() * s64 / (1 - p), (C,), w64, b64, norm.eps)
    ref.backward(gy.double())
    assert float((y1.detach().double() - ref.detach()).abs().max()) <= 5e-6
    for g_, r_ in zip(got, (x64.grad, s64.grad, w64.grad, b64.grad)):
        assert float((g_.double() - r_).abs().max()) <= 2e-5 * max(1.0, float(r_.abs().max()))
    # same site, same step: the same mask (the backward relies on it); another site: another mask; next step: another mask
    _, k1b = run(drop_a)
    assert torch.equal(k1, k1b)
    _, k2 = run(drop_b)
    assert 0.15 < float((k1 != k2).float().mean()) < 0.21                    # independent masks differ on 2 p (1 - p) = 0.18
    TL.advance(DEV)
    _, k3 = run(drop_a)
    assert 0.15 < float((k1 != k3).float().mean()) < 0.21
    # the generator seed matters
    torch.manual_seed(12)
    _, k4 = run(drop_a)
    assert 0.15 < float((k3 != k4).float().mean()) < 0.21


def test_fallbacks_and_switch(monkeypatch):
    from gvl_amd import train_layers as TL
    norm, drop = torch.nn.LayerNorm(64).to(DEV), torch.nn.Dropout(0.0)
    x, sub = _rand(2, 5, 64, seed=1).requires_grad_(True), _rand(2, 5, 64, seed=2)
    assert TL.eligible(x, sub, norm, drop)
    with torch.no_grad():
        assert not TL.eligible(x, sub, norm, drop)                           # inference: gvl_amd/layers.py serves it
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert not TL.eligible(x, sub, norm, drop)
    assert not TL.eligible(x, sub.half(), norm, drop)
    monkeypatch.setenv("GVL_TRAIN_LAYERS", "torch")
    assert not TL.eligible(x, sub, norm, drop)
    y = TL.residual_dropout_norm(x, sub, drop, norm)
    assert float((y - norm(x + sub)).abs().max()) == 0.0


@pytest.mark.parametrize("p,train", [(0.1, True), (0.1, False), (0.0, True)])
def test_relu_dropout_forward_and_maskless_backward(p, train):
    from gvl_amd import train_layers as TL
    from gvl_amd.linear import Linear
    lin = Linear(512, 2048).to(DEV)
    drop = torch.nn.Dropout(p).train(train)
    x = _rand(16, 300, 512, seed=1).requires_grad_(True)
    torch.manual_seed(5)
    TL.advance(DEV)
    h = lin(x)
    pre = h.detach().clone()
    y = TL.relu_dropout(h, torch.nn.functional.relu, drop)
    assert type(y.grad_fn).__name__ == "ReluDropoutBackward"
    pe = p if train else 0.0
    keep = (y.detach() != 0) | (pre <= 0)                                     # where relu passed, the mask is readable
    want = torch.relu(pre) * keep / (1 - pe)
    assert float((y.detach() - want).abs().max()) <= 1e-6
    if pe > 0:
        kept = float(((y.detach() != 0) & (pre > 0)).sum()) / float((pre > 0).sum())
        assert abs(kept - (1 - pe)) < 2e-3
    else:
        assert torch.equal(y.detach(), torch.relu(pre))
    gy = _rand(16, 300, 2048, seed=2)
    y.backward(gy)
    gx, gw, gb = x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()
    x.grad = lin.weight.grad = lin.bias.grad = None
    dh = (gy * ((pre > 0) & keep) / (1 - pe))
    ref_h = torch.nn.functional.linear(x, lin.weight, lin.bias)
    ref_h.backward(dh)
    for a_, b_ in ((gx, x.grad), (gw, lin.weight.grad), (gb, lin.bias.grad)):
        assert float((a_ - b_).abs().max()) <= 1e-4 * max(1.0, float(b_.abs().max()))


def test_relu_dropout_falls_back_for_other_activations():
    from gvl_amd import train_layers as TL
    drop = torch.nn.Dropout(0.0)
    x = _rand(4, 8, 64, seed=1).requires_grad_(True)
    y = TL.relu_dropout(x * 2.0, torch.nn.functional.gelu, drop)
    assert type(y.grad_fn).__name__ != "ReluDropoutBackward" and torch.equal(y, torch.nn.functional.gelu(x * 2.0))
    with torch.no_grad():
        assert TL.relu_dropout(x, torch.nn.functional.relu, drop).grad_fn is None


def test_backward_redraws_the_masks_of_its_own_forward_after_a_later_forward():
    """the backward regenerated its mask from the LIVE step counter; a second training forward between a forward and
    its backward (two micro-batches summed before .backward(), a checkpoint recompute) then masked dsub with another keep set"""
    from gvl_amd import train_layers as TL
    B, Q, C, p = 4, 300, 512, 0.3
    norm = torch.nn.LayerNorm(C).to(DEV)
    drop = torch.nn.Dropout(p).train()
    x = _rand(B, Q, C, seed=3).requires_grad_(True)
    sub = (_rand(B, Q, C, seed=4).abs() + 0.5).requires_grad_(True)
    TL.advance(DEV)
    y1 = TL.residual_dropout_norm(x, sub, drop, norm)
    keep1 = (y1.grad_fn.saved_tensors[0] - x.detach()).abs() > 1e-6
    TL.advance(DEV)                                                          # the next forward begins ...
    y2 = TL.residual_dropout_norm(x.detach(), sub.detach(), drop, norm)      # ... and draws other masks
    keep2 = (y2 - torch.nn.functional.layer_norm(x.detach() + 0, (C,), norm.weight, norm.bias, norm.eps)).abs() > 0
    assert keep2.any()
    y1.backward(_rand(B, Q, C, seed=5))
    passed = sub.grad != 0
    assert torch.equal(passed | ~keep1, torch.ones_like(keep1)) or float((passed ^ keep1).float().mean()) < 1e-4
    assert float((passed & ~keep1).float().sum()) == 0                       # nothing flows through an element forward dropped


def test_row_maxima_ride_on_the_results_forward_and_backward():
    """the kernels leave max |row| of their results for the next Linear product's split (gvl_amd.layers.tag_amax): exact for
    y, y + pos, relu-dropout's output; an upper bound within 1 / (1 - p) for dz / dsub -- and the tags survive autograd's hand-over"""
    from gvl_amd import layers as L
    from gvl_amd import train_layers as TL
    B, Q, C, p = 16, 188, 512, 0.1
    norm = torch.nn.LayerNorm(C).to(DEV)
    drop = torch.nn.Dropout(p).train()
    x, sub = _rand(B, Q, C, seed=3).requires_grad_(True), _rand(B, Q, C, seed=4).requires_grad_(True)
    pos = _rand(Q, C, seed=6).unsqueeze(0).expand(B, -1, -1)
    TL.advance(DEV)
    y = TL.residual_dropout_norm(x, sub, drop, norm, pos)
    assert torch.equal(L.amax_of(y, B * Q), y.detach().abs().amax(-1).reshape(-1))
    q = TL.add_pos(y, pos)
    assert torch.allclose(L.amax_of(q, B * Q), q.detach().abs().amax(-1).reshape(-1), rtol=1e-6, atol=0)
    h = TL.relu_dropout(y, torch.relu, drop)
    assert torch.equal(L.amax_of(h, B * Q), h.detach().abs().amax(-1).reshape(-1))
    seen = {}
    sub.register_hook(lambda g: seen.__setitem__("dsub", (g, L.amax_of(g, B * Q))))
    y.register_hook(lambda g: seen.__setitem__("dy", (g, L.amax_of(g, B * Q))))
    h.backward(_rand(B, Q, C, seed=7))
    for k in ("dsub", "dy"):
        g, am = seen[k]
        assert am is not None, k + ": the tag did not survive"
        true = g.abs().amax(-1).reshape(-1)
        assert bool((am >= true * (1 - 1e-6)).all()) and bool((am <= true / (1 - p) * (1 + 1e-6) + 1e-30).all() or k == "dy")
    y.add_(1.0)                                                              # an in-place edit invalidates the tag
    assert L.amax_of(y, B * Q) is None


@pytest.mark.parametrize("RD", [1, 2])
def test_box_refine_train_node_equals_the_torch_formulation(RD):
    """sigmoid(delta + inverse_sigmoid(ref)) (deformable_transformer.py:314-324, misc/detr_utils/misc.py:582-586) as one node on
    gvl_box_refine_f32 / gvl_box_refine_backward_f32: values, the next layer's scaled reference points and both gradients
    against the PyTorch expression -- reference points at 0, 1, outside [0, 1] and within eps of the ends included (the clamps'
    gradient rules)"""
    from gvl_amd import layers as L
    from gvl_amd.deformable_transformer import inverse_sigmoid
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(RD)
    B, Q, Lv = 3, 50, 4
    delta = (torch.randn(B, Q, 2, device=dev, generator=g) * 2).requires_grad_()
    ref = torch.rand(B, Q, RD, device=dev, generator=g)
    ref[0, :8, 0] = torch.tensor([0.0, 1.0, -0.2, 1.3, 5e-6, 1 - 5e-6, 1e-5, 0.5], device=dev)
    ref.requires_grad_()
    vr = torch.rand(B, Lv, device=dev, generator=g) * 0.5 + 0.5
    assert L.box_refine_train_eligible(delta, ref)
    new_ref, ref_in = L.box_refine_train(delta, ref, vr, True)
    d2, r2 = delta.detach().clone().requires_grad_(), ref.detach().clone().requires_grad_()
    prior = inverse_sigmoid(r2)
    want = (d2 + prior).sigmoid() if RD == 2 else torch.cat([d2[..., :1] + prior, d2[..., 1:]], -1).sigmoid()
    want_in = want.detach()[:, :, None] * torch.stack([vr] * 2, -1)[:, None]
    assert float((new_ref - want).abs().max()) <= 2e-6 and float((ref_in - want_in).abs().max()) <= 2e-6
    go = torch.randn(B, Q, 2, device=dev, generator=g)
    new_ref.backward(go)
    want.backward(go)
    assert float((delta.grad - d2.grad).abs().max()) <= 1e-6 * max(1.0, float(d2.grad.abs().max()))
    assert float((ref.grad - r2.grad).abs().max()) <= 1e-5 * max(1.0, float(r2.grad.abs().max()))


@pytest.mark.parametrize("B,Q,C", [(16, 300, 512), (3, 7, 64), (2, 129, 1028), (1, 1, 4), (5, 33, 260)])
def test_count_pool_train_node_is_torch_max_over_the_queries(B, Q, C):
    """predict_event_num's pooling (pdvc.py:317 `torch.max(hs_lid, dim=1)`) as one node on gvl_count_pool_f32 /
    gvl_count_pool_backward_f32: the values bit for bit, the gradient on the selected row of every column alone -- with repeated
    maxima in some columns (the first such row takes it, the rows' gradients still sum to the incoming one)"""
    from gvl_amd import layers as L
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(B * 1000 + Q)
    hs = torch.randn(B, Q, C, device=dev, generator=g)
    if Q > 4:
        hs[0, 3, :4] = 9.0
        hs[0, Q - 1, :4] = 9.0                               # ties: rows 3 and Q - 1
    hs.requires_grad_()
    assert L.count_pool_train_eligible(hs)
    pooled = L.count_pool_train(hs)
    want_v, want_i = hs.detach().max(dim=1)
    assert torch.equal(pooled, want_v)
    go = torch.randn(B, C, device=dev, generator=g)
    pooled.backward(go)
    first = (hs.detach() == want_v[:, None, :]).int().argmax(dim=1)          # first row attaining the maximum
    want_g = torch.zeros_like(hs).scatter_(1, first[:, None, :], go[:, None, :])
    assert torch.equal(hs.grad, want_g)
    if Q > 4:
        assert int(first[0, 0]) == 3


@pytest.mark.parametrize("B,Q,C,parts", [(16, 300, 512, 2), (3, 7, 8, 2), (1, 5, 4, 1), (9, 33, 36, 3)])
def test_expand_parts_node_is_chunk_expand_with_a_one_launch_gradient(B, Q, C, parts):
    """query_pos, tgt = chunk(query_embed), each .expand(bs, -1, -1) (deformable_transformer.py:128-135) as one node: the same
    stride-0 views forward, the embedding's gradient = the blocks' gradients summed over the batch (gvl_batch_sum_f32), also when
    no gradient reaches one block"""
    from gvl_amd import layers as L
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(B + Q)
    emb = torch.randn(Q, parts * C, device=dev, generator=g, requires_grad=True)
    assert L.expand_parts_eligible(emb, parts)
    outs = L.expand_parts(emb, B, parts)
    ref = emb.detach().clone().requires_grad_()
    want = [c.unsqueeze(0).expand(B, -1, -1) for c in torch.chunk(ref, parts, dim=1)]
    for o, w_ in zip(outs, want):
        assert o.shape == w_.shape and o.stride() == w_.stride() and torch.equal(o, w_)
    ws = [torch.randn(B, Q, C, device=dev, generator=g) for _ in range(parts)]
    sum((o * w_).sum() for o, w_ in zip(outs, ws)).backward()
    sum((o * w_).sum() for o, w_ in zip(want, ws)).backward()
    assert float((emb.grad - ref.grad).abs().max()) <= 1e-5 * float(ref.grad.abs().max())
    # rows=True: the un-expanded blocks too; their gradients land in the same embedding gradient
    emb.grad, ref.grad = None, None
    outs = L.expand_parts(emb, B, parts, rows=True)
    assert len(outs) == 2 * parts and all(torch.equal(outs[parts + h], emb.detach()[:, h * C:(h + 1) * C]) for h in range(parts))
    wr = torch.randn(Q, C, device=dev, generator=g)
    ((outs[0] * ws[0]).sum() + (outs[parts] * wr).sum()).backward()
    ((want[0] * ws[0]).sum() + (ref[:, :C] * wr).sum()).backward()
    assert float((emb.grad - ref.grad).abs().max()) <= 1e-5 * float(ref.grad.abs().max())
    emb.grad = None
    (L.expand_parts(emb, B, parts, rows=True)[parts] * wr).sum().backward()          # the rows alone
    assert torch.equal(emb.grad[:, :C], wr) and float(emb.grad[:, C:].abs().sum()) == 0.0
    if parts > 1:                                             # only the last block is used
        emb.grad = None
        L.expand_parts(emb, B, parts)[-1].mul(ws[-1]).sum().backward()
        assert float(emb.grad[:, :(parts - 1) * C].abs().max()) == 0.0
        assert float((emb.grad[:, (parts - 1) * C:] - ws[-1].sum(0)).abs().max()) <= 1e-5 * float(ws[-1].sum(0).abs().max())


@pytest.mark.parametrize("B,Q,C", [(16, 300, 512), (2, 9, 64)])
def test_class_and_count_heads_node_equals_linear_and_max(B, Q, C):
    """class head (Linear(C, 1), pdvc.py:455) + the count head's pooling (pdvc.py:317) as one node: values and every gradient
    against the PyTorch formulation, also when only one of the two outputs is used"""
    from gvl_amd import layers as L
    dev = torch.device("cuda:0")
    torch.manual_seed(B)
    head = torch.nn.Linear(C, 1).to(dev)
    hs = torch.randn(B, Q, C, device=dev, requires_grad=True)
    assert L.class_count_heads_eligible(head, hs)
    for use in ((True, True), (True, False), (False, True)):
        hs.grad = None
        head.zero_grad(set_to_none=True)
        logits, pooled = L.class_count_heads(head, hs)
        ref_h = hs.detach().clone().requires_grad_()
        ref_head = torch.nn.Linear(C, 1).to(dev)
        ref_head.load_state_dict(head.state_dict())
        want_l, want_p = ref_head(ref_h), ref_h.max(dim=1)[0]
        assert float((logits - want_l).abs().max()) <= 1e-5 and torch.equal(pooled, want_p)
        gl, gp = torch.randn_like(want_l), torch.randn_like(want_p)
        loss = (logits * gl).sum() * use[0] + (pooled * gp).sum() * use[1]
        want = (want_l * gl).sum() * use[0] + (want_p * gp).sum() * use[1]
        loss.backward()
        want.backward()
        assert float((hs.grad - ref_h.grad).abs().max()) <= 1e-5 * max(1.0, float(ref_h.grad.abs().max()))
        if use[0]:
            assert float((head.weight.grad - ref_head.weight.grad).abs().max()) <= 2e-4 * float(ref_head.weight.grad.abs().max())
            assert float((head.bias.grad - ref_head.bias.grad).abs().max()) <= 2e-4 * max(1.0, float(ref_head.bias.grad.abs().max()))


@pytest.mark.parametrize("fan", [2, 3, 6])
def test_fan_out_handles_sum_their_gradients_inside_the_backward_kernel(fan):
    """residual_dropout_norm(fan=k): k handles of one result (one per consumer); the backward kernel adds the gradients that
    arrive on them as it loads them -- the same input gradients as one handle consumed k times (autograd's own adds), also when
    one handle stays unused"""
    from gvl_amd import train_layers as TL
    torch.manual_seed(fan)
    B, Q, C = 4, 150, 512
    norm = torch.nn.LayerNorm(C).to(DEV)
    drop = torch.nn.Dropout(0.0)
    x0, s0 = _rand(B, Q, C, seed=1), _rand(B, Q, C, seed=2)
    ws = [_rand(B, Q, C, seed=10 + i) for i in range(fan)]
    for used in (fan, fan - 1):
        grads = []
        for mode in ("fan", "plain"):
            x, s = x0.clone().requires_grad_(), s0.clone().requires_grad_()
            norm.zero_grad(set_to_none=True)
            if mode == "fan":
                ys = TL.residual_dropout_norm(x, s, drop, norm, fan=fan)
                assert len(ys) == fan and all(y.data_ptr() == ys[0].data_ptr() for y in ys)
            else:
                ys = (TL.residual_dropout_norm(x, s, drop, norm),) * fan
            sum((y * w_).sum() for y, w_ in list(zip(ys, ws))[:used]).backward()
            grads.append((x.grad, s.grad, norm.weight.grad.clone(), norm.bias.grad.clone()))
        for a, b in zip(*grads):
            assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("B,C,lengths", [(16, 512, [100, 50, 25, 13]), (3, 64, [7, 1]), (2, 260, [5, 4, 3, 2, 1])])
def test_level_pos_embed_node_equals_the_torch_formulation(B, C, lengths):
    """cat_l(pos_l^T + level_embed[l]) (deformable_transformer.py:96-103) as one node: same values, the level embedding's gradient
    (per-level sums over videos and rows, gvl_level_sums_f32 + gvl_batch_sum_f32) against autograd's"""
    from gvl_amd import layers as L
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(B + C)
    emb = torch.randn(len(lengths) + 1, C, device=dev, generator=g, requires_grad=True)   # (one level more than used)
    pos = [torch.randn(B, C, T, device=dev, generator=g).requires_grad_(l % 2 == 0) for l, T in enumerate(lengths)]
    assert L.level_pos_embed_eligible(emb, pos)
    out = L.level_pos_embed(emb, pos)
    ref = emb.detach().clone().requires_grad_()
    pos_ref = [p.detach().clone().requires_grad_(p.requires_grad) for p in pos]
    want = torch.cat([p.transpose(1, 2) + ref[l].view(1, 1, -1) for l, p in enumerate(pos_ref)], 1)
    assert torch.equal(out, want)
    go = torch.randn(B, sum(lengths), C, device=dev, generator=g)
    out.backward(go)
    want.backward(go)
    assert float((emb.grad - ref.grad).abs().max()) <= 1e-5 * max(1.0, float(ref.grad.abs().max()))
    assert float(emb.grad[len(lengths)].abs().max()) == 0.0
    for p, r in zip(pos, pos_ref):                            # (the learned half of the position embedding: its slice of the gradient)
        assert (p.grad is None) == (r.grad is None) and (p.grad is None or torch.equal(p.grad, r.grad))


def test_mask_rows_node_is_masked_fill_in_place_with_row_maxima_backward():
    """gvl_amd.layers.mask_rows on a Linear's fresh output: the padded rows zeroed in place (the tensor returned IS the Linear's),
    the gradient with the same rows zeroed and tagged with its row maxima; leaves / views / other dtypes take masked_fill"""
    from gvl_amd import layers as L
    from gvl_amd import linear as GL
    torch.manual_seed(5)
    B, S, C = 16, 188, 512
    lin = GL.Linear(C, C).to(DEV)
    x = torch.randn(B, S, C, device=DEV, requires_grad=True)
    mask = torch.zeros(B, S, dtype=torch.bool, device=DEV)
    for b in range(B):
        mask[b, S - 3 * b:] = True                              # ragged padding, video 0 unpadded
    y = lin(x)
    ptr = y.data_ptr()
    out = L.mask_rows(y, mask)
    assert out.data_ptr() == ptr
    ref_x = x.detach().clone().requires_grad_()
    ref_lin = GL.Linear(C, C).to(DEV)
    ref_lin.load_state_dict(lin.state_dict())
    want = ref_lin(ref_x).masked_fill(mask[..., None], 0.0)
    assert torch.equal(out, want)
    go = torch.randn(B, S, C, device=DEV)
    out.backward(go)
    want.backward(go)
    assert float((x.grad - ref_x.grad).abs().max()) <= 2e-6 * float(ref_x.grad.abs().max())
    assert float((lin.weight.grad - ref_lin.weight.grad).abs().max()) <= 2e-6 * float(ref_lin.weight.grad.abs().max())
    # the node's own backward: zeroed rows + row maxima
    g = torch.randn(B, S, C, device=DEV)
    dx, _ = L._MaskRows.backward(type("C", (), {"saved_tensors": (mask.view(torch.uint8),)})(), g)
    assert torch.equal(dx, g.masked_fill(mask[..., None], 0.0))
    assert torch.equal(L.amax_of(dx, B * S), dx.abs().amax(-1).flatten())
    leaf = torch.randn(B, S, C, device=DEV, requires_grad=True)
    assert L.mask_rows(leaf, mask).data_ptr() != leaf.data_ptr()                  # (a leaf: out of place)


def test_embed_rows_node_is_index_select_with_a_gradient_that_skips_zero_rows():
    """gvl_amd.layers.embed_rows: the captioner's embedding lookup; its backward (gvl_index_add_rows_f32) equals index_add -- with
    half of the gradient rows zero and pointing at one entry (the padded positions), repeated indices, an odd width"""
    from gvl_amd import layers as L
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    for V, E, n in ((8518, 512, 4416), (50, 68, 300), (7, 1028, 5)):
        w = torch.randn(V, E, device=dev, generator=g, requires_grad=True)
        ids = torch.randint(0, V, (n,), device=dev, generator=g)
        ids[: n // 2] = 0                                         # <pad>
        ids[n // 2: n // 2 + n // 8] = 1                          # a hot word
        out = L.embed_rows(w, ids)
        assert torch.equal(out, w.detach()[ids])
        go = torch.randn(n, E, device=dev, generator=g)
        go[: n // 2] = 0.0
        go[-1, ::2] = 0.0                                         # (zeros inside a live row are skipped element-wise)
        out.backward(go)
        want = torch.zeros(V, E, device=dev, dtype=torch.float64).index_add_(0, ids, go.double())
        assert float((w.grad.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))


def test_captioner_slab_node_equals_project_value_ctx2att_cat(monkeypatch):
    """gvl_amd.CaptioningHead.LSTM_DSA._CapSlab -- [value_proj(memory) masked | ctx2att(.)] with both halves written in place and a
    hand-written backward -- against the formulation it replaces (project_value, ctx2att, cat), at the cfg A memory shape: the slab,
    the memory gradient and the four parameter gradients"""
    from gvl_amd.config import make_opt
    from gvl_amd.CaptioningHead.LSTM_DSA import ShowAttendTellCore
    torch.manual_seed(7)
    opt = make_opt("anet_tsp_ssvg", device="cuda")
    core = ShowAttendTellCore(opt).to(DEV).train()
    with torch.no_grad():
        core.deformable_att.value_proj.bias.normal_(0, 0.1)
        core.ctx2att.bias.normal_(0, 0.1)
    B, S, C = 16, 188, 512
    mem0 = torch.randn(B, S, C, device=DEV)
    mask = torch.zeros(B, S, dtype=torch.bool, device=DEV)
    for b in range(B):
        mask[b, S - 2 * b:] = True
    query = torch.randn(B, 12, C * 2 if core.deformable_att.sampling_offsets.weight.shape[1] == core.rnn_size + 2 * C else C, device=DEV)
    query = query[..., :core.deformable_att.sampling_offsets.weight.shape[1] - core.rnn_size]
    go = torch.randn(B, S, 1, 1024, device=DEV)
    from gvl_amd.CaptioningHead.LSTM_DSA import _cap_slab_eligible
    assert _cap_slab_eligible(core, mem0.clone().requires_grad_(), mask)
    res = {}
    ps = [core.deformable_att.value_proj.weight, core.deformable_att.value_proj.bias, core.ctx2att.weight, core.ctx2att.bias]
    for mode in ("node", "torch"):
        if mode == "torch":
            monkeypatch.setenv("GVL_CAP_SLAB", "torch")
        mem = mem0.clone().requires_grad_()
        for p_ in core.parameters():
            p_.grad = None
        slab = core.prepare(query, mem, mask)["slab"]
        (slab * go).sum().backward()
        res[mode] = [slab.detach().clone(), mem.grad.clone()] + [p_.grad.clone() for p_ in ps]
    for a, b in zip(res["node"], res["torch"]):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())), (a.shape, float((a - b).abs().max()))
    assert float(res["node"][0][0, S - 1, 0, :512].abs().max()) > 0.0              # (video 0 has no padded row)
    assert float(res["node"][0][5, S - 1, 0, :512].abs().max()) == 0.0              # a padded row: the value half is zero
