"""Training under torch.autocast (BASELINE config 5 names bf16; the reference has only the flag, pdvc.py:214-215) on the hand-written
path: gvl_amd.pdvc.autocast_training_policy -- default "f16": the fp32-storage training kernels with ONE fp16 matrix-core product
per fp32 product in the forward, input-gradient and weight-gradient products (11-bit operands at their row / tensor scale, fp32
accumulation, fp32 master weights).  Checked here:
  * the same own kernels serve the autocast step and the fp32 step (path census), and the backward's products really run at one
    product (the count is recorded per autograd node: the backward runs on autograd's thread, after the forward's context ended);
  * every loss term and every parameter's gradient norm stay within the single-product error model of the fp32 step's;
  * GVL_AUTOCAST_TRAINING=bf16 still selects torch's own autocast formulation."""
import pytest
import torch

from helpers import load, path_census
from test_gpu_full_dims import build_anet, train_batch, location_fed

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KW = dict(transformer_dropout_prob=0.0, drop_prob=0.0)


def _step(model, criterion, dt, autocast):
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        out, loss = model(dt, criterion, None, "queries")
    wd = criterion.weight_dict
    final = sum(loss[k].float() * wd[k] for k in loss.keys() if k in wd)
    final.backward()
    return ({k: float(v) for k, v in loss.items()}, float(final),
            {n: p_.grad.detach().clone() for n, p_ in model.named_parameters() if p_.grad is not None})


def test_autocast_train_step_stays_on_the_hand_written_path_and_within_the_one_product_error():
    f, opt, model_a, crit_a = build_anet(True, **KW)
    _, _, model_b, crit_b = build_anet(True, **KW)
    dt = train_batch(f, load("pdvc_anet_full_train"))
    (loss_a, final_a, grads_a), tags_a = path_census(lambda: _step(model_a, crit_a, dt, False))
    (loss_b, final_b, grads_b), tags_b = path_census(lambda: _step(model_b, crit_b, dt, True))
    # the same kernels, the same number of times: autocast left none of the layers to the library
    assert tags_a == tags_b and sum(tags_a.values()) > 100, (tags_a - tags_b, tags_b - tags_a)
    assert all(g.dtype == torch.float32 for g in grads_b.values())
    # operands rounded to 11 significant bits (relative 2^-12 each, independent signs): a product over K = 512 terms carries ~3e-4 /
    # sqrt(K) of relative noise per layer, the step's ~40 products in sequence and the matcher's discrete choice on top
    assert abs(final_b - final_a) <= 2e-3 * abs(final_a), (final_a, final_b)
    for k in loss_a:
        assert abs(loss_b[k] - loss_a[k]) <= 2e-3 * max(1.0, abs(loss_a[k])), (k, loss_a[k], loss_b[k])
    worst, diff = {}, {}
    assert grads_a.keys() == grads_b.keys()
    for n in grads_a:
        na, nb = float(grads_a[n].norm()), float(grads_b[n].norm())
        rel = abs(na - nb) / max(1e-3, na)
        worst[n] = rel
        if location_fed(n):
            # sums of sampling-LOCATION gradients: piecewise constant in the location with heavy cancellation (the fp32 goldens
            # already move by ~1 % under a 1e-6 perturbation, tests/test_gpu_full_dims.py); a 2^-12 rounding of the offsets'
            # operands moves samples across frame boundaries -- same magnitude, not the same value
            assert torch.isfinite(grads_b[n]).all() and 0.2 * na <= nb <= 5.0 * na, (n, na, nb)
        else:
            assert rel <= 1e-2, (n, na, nb)
        diff[n] = float((grads_a[n] - grads_b[n]).norm()) / max(1e-3, na)
    print("largest gradient-norm deviations:", sorted(worst.items(), key=lambda kv: -kv[1])[:5])
    print("largest relative gradient differences:", sorted(diff.items(), key=lambda kv: -kv[1])[:8])
    # element-wise (relative to the gradient's norm): gradients downstream of sampling LOCATIONS are piecewise in them (a sample
    # that crosses a frame boundary under the rounding switches rows), the others follow the products' noise
    for n, d in diff.items():
        if not location_fed(n):
            assert d <= (3e-1 if "attention_weights" in n else 1e-1), (n, d)
    assert any(not torch.equal(grads_a[n], grads_b[n]) for n in grads_a)          # (it did run at lower precision)


def test_one_product_policy_is_closer_to_fp32_than_torchs_bf16_autocast(monkeypatch):
    """the default policy lowers precision LESS than the formulation autocast would run (11-bit against 8-bit operands): over the
    parameters whose gradient does not collect sampling-location gradients, the median relative distance to the fp32 step's
    gradient is smaller than that of GVL_AUTOCAST_TRAINING=bf16"""
    import statistics
    f, opt, model_a, crit_a = build_anet(True, **KW)
    dt = train_batch(f, load("pdvc_anet_full_train"))
    _, _, ga = _step(model_a, crit_a, dt, False)
    dist = {}
    for pol in ("f16", "bf16"):
        monkeypatch.setenv("GVL_AUTOCAST_TRAINING", pol)
        _, _, model_b, crit_b = build_anet(True, **KW)
        _, _, gb = _step(model_b, crit_b, dt, True)
        dist[pol] = [float((ga[n] - gb[n].float()).norm()) / max(1e-3, float(ga[n].norm())) for n in ga if not location_fed(n)]
    m16, mbf = statistics.median(dist["f16"]), statistics.median(dist["bf16"])
    print("median relative gradient distance to the fp32 step: one fp16 product", m16, "| torch bf16 autocast", mbf)
    assert m16 < mbf


def test_backward_products_follow_the_forward():
    """_TrainLinearFunction under MSDA.f16_products(1): dx and dW / db come from ONE product too (recorded on the node), i.e. they
    differ from the three-product gradients by the 11-bit rounding -- and from nothing else"""
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    from gvl_amd import linear as GL
    torch.manual_seed(0)
    lin = torch.nn.Linear(512, 1024).to(DEV)
    x = torch.randn(2048, 512, device=DEV)
    g = torch.randn(2048, 1024, device=DEV)
    res = {}
    for n in (3, 1):
        xi = x.clone().requires_grad_()
        lin.zero_grad(set_to_none=True)
        with MSDA.f16_products(n):
            assert GL.train_linear_eligible(xi, (lin.weight,), (lin.bias,))
            y = GL.train_linear(xi, (lin.weight,), (lin.bias,))
        assert MSDA.f16_products_now() == 3
        y.backward(g)                                               # outside the context, on autograd's thread
        res[n] = (y.detach(), xi.grad, lin.weight.grad.clone(), lin.bias.grad.clone())
    x64, w64, g64 = x.double(), lin.weight.detach().double(), g.double()
    exact = (x64 @ w64.t() + lin.bias.detach().double(), g64 @ w64, g64.t() @ x64, g64.sum(0))
    for i, name in enumerate(("y", "dx", "dW", "db")):
        e3 = float((res[3][i].double() - exact[i]).abs().max() / exact[i].abs().max())
        e1 = float((res[1][i].double() - exact[i]).abs().max() / exact[i].abs().max())
        assert e3 <= 2e-6, (name, e3)
        if name != "db":                                            # (the bias gradient is an fp32 column sum in both)
            assert 2e-6 < e1 <= 1e-3, (name, e1, e3)                # one product: 11-bit operands, nothing worse


def test_bf16_policy_still_selects_torchs_autocast(monkeypatch):
    monkeypatch.setenv("GVL_AUTOCAST_TRAINING", "bf16")
    f, opt, model, crit = build_anet(True, **KW)
    dt = train_batch(f, load("pdvc_anet_full_train"))
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out, loss = model(dt, crit, None, "queries")
    assert out["pred_logits"].dtype == torch.bfloat16               # the heads' Linear ran as a bf16 library GEMM
    assert all(torch.isfinite(v).all() for v in loss.values() if isinstance(v, torch.Tensor))
