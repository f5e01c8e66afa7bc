"""train -> eval -> train -> eval in ONE process (the reference's train.py evaluates after every epoch, train.py:411-470): the
update kernels (gvl_clip_adam_step_f32) and the hipGraph replays of a captured step write parameters through raw pointers, and
every weight-derived cache of gvl_amd -- split-fp16 operand planes, the captioner's stacked matrices, the captured eval / decode
graphs -- is keyed on (data_ptr, _version).  After training, an evaluation must see the NEW weights: it is compared with a fresh
model loaded from the state_dict."""
import pytest
import torch

from helpers import load, maxerr
from test_gpu_full_dims import build_anet, train_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KEYS = ("pred_logits", "pred_boxes", "pred_count")


def _eval(model, criterion, dt, graphed=None):
    model.eval()
    with torch.no_grad():
        out, _ = graphed(dt) if graphed is not None else model(dt, criterion, None, "queries", eval_mode=True)
    res = {k: out[k].detach().clone() for k in KEYS}
    # (a model whose every caption ends at once returns seq = [], LSTM_DSA.py:186-187)
    res["seq"] = out["seq"].detach().clone() if isinstance(out["seq"], torch.Tensor) else torch.zeros(0, device=DEV)
    return res


def _fresh_eval(model, dt, kw):
    _, _, fresh, crit = build_anet(False, **kw)
    fresh.load_state_dict(model.state_dict(), strict=True)
    return _eval(fresh, crit, dt)


def _same(a, b, what):
    for k in KEYS:
        assert maxerr(a[k], b[k]) <= 2e-5 * max(1.0, float(b[k].abs().max())), (what, k, maxerr(a[k], b[k]))
    assert a["seq"].shape == b["seq"].shape, (what, a["seq"].shape, b["seq"].shape)
    assert a["seq"].numel() == 0 or float((a["seq"] == b["seq"]).float().mean()) >= 0.999, what


def _moved(a, b):
    return maxerr(a["pred_logits"], b["pred_logits"]) > 1e-3


@pytest.mark.parametrize("graphed_train", [False, True])
def test_evaluation_after_training_runs_on_the_updated_weights(graphed_train):
    from gvl_amd.parallel import GraphedEvalForward, GraphedTrainStep, TrainStep
    kw = dict(transformer_dropout_prob=0.0, drop_prob=0.0, lr=5e-5, weight_decay=1e-4, grad_clip=100.0)
    f, opt, model, criterion = build_anet(True, **kw)
    dt = train_batch(f, load("pdvc_anet_full_train"))
    graphed_eval = GraphedEvalForward(model, criterion)
    e0, g0 = _eval(model, criterion, dt), _eval(model, criterion, dt, graphed_eval)
    _same(g0, e0, "graphed eval, initial weights")
    step = (GraphedTrainStep(model, criterion, opt, max_gt=10, max_cap_len=20, max_events=40) if graphed_train
            else TrainStep(model, criterion, opt, capturable=True))
    prev = e0
    for round_ in range(2):                                  # train -> eval -> train -> eval
        model.train()
        for _ in range(3):
            step(dt)
        if not graphed_train:
            assert float(step.optimizer.state[step.params[0]]["step"]) == 3.0 * (round_ + 1)
        e1, g1 = _eval(model, criterion, dt), _eval(model, criterion, dt, graphed_eval)
        ref = _fresh_eval(model, dt, kw)
        assert _moved(ref, prev), "the training steps did not move the outputs: the test would not see a stale cache"
        _same(e1, ref, f"eager eval after training round {round_}")
        _same(g1, ref, f"graphed eval after training round {round_}")
        prev = ref
