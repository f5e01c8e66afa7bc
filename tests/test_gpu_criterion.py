"""Fused matcher-cost / set-criterion kernels (gvl_criterion.hip) against the PyTorch op sequences that mirror the
reference (gvl_amd.matcher.HungarianMatcher.cost_matrix, gvl_amd.criterion.SetCriterion.loss_* -- themselves pinned to
the reference by the eval / train goldens in test_gpu_model.py)."""
import argparse

import pytest
import torch

pytestmark = pytest.mark.gpu


def make(B, Q, NC, sizes, nl, seed, dev, degenerate=False):
    g = torch.Generator().manual_seed(seed)
    layers = []
    for _ in range(nl):
        boxes = torch.stack([torch.rand(B, Q, generator=g), torch.rand(B, Q, generator=g) * 0.5 + 0.01], -1)
        layers.append({"pred_logits": torch.randn(B, Q, NC, generator=g).to(dev).requires_grad_(),
                       "pred_count": torch.randn(B, 11, generator=g).to(dev).requires_grad_(),
                       "pred_boxes": boxes.to(dev).requires_grad_()})
    targets = []
    for n in sizes:
        tb = torch.stack([torch.rand(n, generator=g) * 0.5 + 0.25, torch.rand(n, generator=g) * 0.3 + 0.1], -1)
        targets.append({"boxes": tb.to(dev), "labels": torch.randint(0, NC, (n,), generator=g).to(dev)})
    if degenerate:
        # ties for the subgradient conventions: a prediction equal to its target, two identical predictions
        with torch.no_grad():
            for o in layers:
                o["pred_boxes"][0, 0] = targets[0]["boxes"][0]
                o["pred_boxes"][0, 1] = targets[0]["boxes"][0]
                o["pred_boxes"][0, 2] = torch.tensor([0.1, 0.05], device=dev)      # disjoint from everything
    return layers, targets


def build_criterion(NC):
    from gvl_amd.criterion import SetCriterion
    from gvl_amd.matcher import HungarianMatcher
    opt = argparse.Namespace(lloss_gau_mask=1, lloss_beta=1, set_cost_caption=0)
    matcher = HungarianMatcher(cost_class=2, cost_bbox=0, cost_giou=4, cost_alpha=0.25, cost_gamma=2, cost_cl=2.0, opt=opt)
    return SetCriterion(NC, matcher, {}, ['labels', 'boxes', 'cardinality'], focal_alpha=0.25, focal_gamma=2, opt=opt)


@pytest.mark.parametrize("NC,w_bbox", [(1, 0.0), (3, 0.0), (1, 1.5)])
def test_match_cost_kernel_equals_torch_sequence(NC, w_bbox):
    dev = torch.device("cuda:0")
    layers, targets = make(4, 37, NC, [3, 1, 5, 2], 2, 1, dev)
    crit = build_criterion(NC)
    m = crit.matcher
    m.cost_bbox = w_bbox
    tgt_cat = (torch.cat([v["labels"] for v in targets]), torch.cat([v["boxes"] for v in targets]))
    Cf, okf = m.cost_matrices(layers, targets, tgt_cat, fused=True)
    Ct, okt = m.cost_matrices(layers, targets, tgt_cat, fused=False)
    assert bool(okf.all()) and bool(okt)
    # same operations in the same order, one rounding each; what remains is the last bit of expf / logf between this
    # library's device libm calls and PyTorch's kernels (the GIoU term alone is bit-identical)
    assert float((Cf - Ct).abs().max()) <= 2e-6
    m.cost_class, keep = 0.0, m.cost_class
    a, _ = m.cost_matrices(layers, targets, tgt_cat, fused=True)
    b, _ = m.cost_matrices(layers, targets, tgt_cat, fused=False)
    m.cost_class = keep
    if w_bbox == 0.0:
        assert torch.equal(a, b)
    layers[1]["pred_boxes"].data[2, 5, 1] = -0.2         # x1 < x0: the reference asserts (box_ops.py:39-40)
    _, okf = m.cost_matrices(layers, targets, tgt_cat, fused=True)
    assert not bool(okf.all())


@pytest.mark.parametrize("NC,sizes,degenerate", [(1, [3, 2, 4, 3], False), (1, [3, 2, 4, 3], True),
                                                 (3, [2, 5, 3], False), (1, [3, 1, 2], False),
                                                 (1, [2, 0, 3], False),            # a video without ground-truth events
                                                 (1, [30, 12], False)])            # gt_proposal_sample_num = 30
def test_fused_criterion_equals_torch_formulation(NC, sizes, degenerate):
    from gvl_amd.criterion import LOSS_KEYS
    dev = torch.device("cuda:0")
    B = len(sizes)
    res = {}
    for fused in (True, False):
        layers, targets = make(B, 41, NC, sizes, 2, 7, dev, degenerate)
        crit = build_criterion(NC)
        crit.fused = fused
        outputs = dict(layers[0])
        outputs["aux_outputs"] = [layers[1]]
        losses, last, aux = crit(outputs, targets)
        assert sorted(losses) == sorted([k for k in LOSS_KEYS] + [k + "_0" for k in LOSS_KEYS])
        g = torch.Generator().manual_seed(3)
        wts = {k: float(torch.rand(1, generator=g)) + 0.5 for k in sorted(losses)}
        total = sum(losses[k] * wts[k] for k in sorted(losses) if "cardinality" not in k and not torch.isnan(losses[k]))
        total.backward()
        res[fused] = ({k: float(v) for k, v in losses.items()},
                      [o[k].grad.clone() for o in layers for k in ("pred_logits", "pred_count", "pred_boxes")])
    for k, v in res[False][0].items():
        f = res[True][0][k]
        assert (v != v and f != f) or abs(f - v) <= 2e-6 * max(1.0, abs(v)), (k, f, v)     # nan == nan (single match)
    for a, b in zip(res[True][1], res[False][1]):
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("R,C,pad", [(4800, 512, 0), (4800, 2048, 0), (1056, 8519, 0), (96, 2576, 0), (1, 4, 0), (65, 7, 3),
                                     (300, 260, 12), (2000, 1, 0)])
def test_col_sum_matches_float64(R, C, pad):
    """gvl_col_sum_f32 (the bias gradient of gvl_amd.linear) on 16-byte-row and odd-width / strided matrices."""
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    g = torch.Generator(device="cpu").manual_seed(R + C)
    full = torch.randn(R, C + pad, generator=g).cuda()
    x = full[:, :C]
    got = MSDA.col_sum(x)
    want = x.double().sum(0)
    assert got.shape == (C,)
    assert (got.double() - want).abs().max().item() <= 1e-5 * max(1.0, R ** 0.5) * 4


def test_linear_gradients_equal_autograd_of_f_linear(monkeypatch):
    """the library-GEMM form of gvl_amd.linear.Linear (GVL_TRAIN_LINEAR=torch: the A/B switch of the hand-written training
    products, tests/test_gpu_train_linear.py): autograd's own two GEMMs + the column-sum kernel for the bias"""
    import torch.nn.functional as F
    from gvl_amd import linear as GL
    from gvl_amd.linear import Linear, linear
    monkeypatch.setattr(GL, "_TRAIN_LINEAR", False)
    torch.manual_seed(3)
    lin = Linear(512, 2048).cuda()
    x = torch.randn(16, 300, 512, device="cuda", requires_grad=True)
    gy = torch.randn(16, 300, 2048, device="cuda")
    y = lin(x)
    assert type(y.grad_fn).__name__ == "_LinearFunctionBackward"
    y.backward(gy)
    got = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = None
    lin.zero_grad()
    F.linear(x, lin.weight, lin.bias).backward(gy)
    assert torch.equal(got[0], x.grad) and torch.equal(got[1], lin.weight.grad)      # the same two GEMM calls
    assert (got[2] - lin.bias.grad).abs().max().item() <= 2e-4 * lin.bias.grad.abs().max().item()
    # not applicable -> plain F.linear: no bias, no grad, autocast
    assert type(linear(x, lin.weight, None).grad_fn).__name__ != "_LinearFunctionBackward"
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert type(lin(x).grad_fn).__name__ != "_LinearFunctionBackward"


@pytest.mark.parametrize("n,steps,H,V", [(192, 23, 512, 8518), (5, 3, 64, 37), (40, 7, 96, 1000)])
def test_vocab_nll_equals_log_softmax_gather_and_its_autograd(n, steps, H, V):
    """gvl_amd.linear.vocab_nll (gvl_ce_rows_*_f32: the masked target log-prob without the log-prob tensor) against the
    reference formulation -- build_loss over F.log_softmax(logit(x)) (LSTM_DSA.py:48-52, 121-123) -- and its autograd:
    values and the gradients of the hidden states, the vocabulary weight and bias; masked rows (padding, steps past the
    caption's end) contribute exactly zero."""
    import torch.nn.functional as F
    from gvl_amd.linear import vocab_nll, vocab_nll_eligible
    g = torch.Generator().manual_seed(n + V)
    x = (torch.randn(n, steps, H, generator=g) * 0.5).cuda().requires_grad_()
    w = (torch.rand(V, H, generator=g) * 0.2 - 0.1).cuda().requires_grad_()
    b = (torch.randn(V, generator=g) * 0.1).cuda().requires_grad_()
    target = torch.randint(0, V, (n, steps), generator=g).cuda()
    lens = torch.randint(0, steps + 1, (n,), generator=g)
    mask = (torch.arange(steps)[None, :] < lens[:, None]).cuda()
    gout = torch.randn(n, generator=g).cuda()
    assert vocab_nll_eligible(x, w, b)
    picked = vocab_nll(x, w, b, target, mask).view(n, steps)
    loss = -picked.sum(1) / (mask.sum(1) + 1e-6)
    (loss * gout).sum().backward()
    got = (loss.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    logp = F.log_softmax(F.linear(x.double(), w.double(), b.double()), dim=2)
    ref = -(logp.gather(2, target[:, :, None]).squeeze(2) * mask).sum(1) / (mask.sum(1) + 1e-6)
    (ref * gout.double()).sum().backward()
    want = (ref.detach(), x.grad, w.grad, b.grad)
    for name, a, r in zip(("loss", "d hidden", "d weight", "d bias"), got, want):
        err = float((a.double() - r.double()).abs().max())
        assert err <= 2e-5 * max(1.0, float(r.abs().max())), (name, err, float(r.abs().max()))
    dead = ~mask.any(1)
    if bool(dead.any()):
        assert float(got[0][dead].abs().max()) == 0.0 and float(got[1][dead].abs().max()) == 0.0
