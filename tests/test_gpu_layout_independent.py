"""Layout-independent captured steps (VERDICT r1 item 1 / ADVICE: graph caches keyed on data-dependent batch layouts).

The reference's loops feed batches whose number of events per video and caption lengths differ every iteration
(eval_utils.py:187-203, train.py:385-398; matcher.py:120-128 splits the cost matrix by those counts).  These tests pin
the padded form (gvl_amd.targets.PaddedTargets: fixed shapes + device-side counts) to the list form the reference uses,
and the graphed steps to the eager steps across batches of DIFFERENT layouts with exactly one captured graph."""
import numpy as np
import pytest
import torch

from helpers import load, pdvc_state, maxerr
from bench import synth_batch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
LAYOUTS = [([3, 2], (2, 3)), ([0, 5], (1, 4)), ([7, 1], (3, 3)), ([1, 1], (2, 2)), ([4, 4], (1, 1)), ([8, 0], (2, 4)),
           ([2, 6], (1, 3)), ([5, 3], (4, 4))]


def toy(train, **over):
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    f = load("pdvc_eval")
    kw = dict(num_queries=8, feature_dim=64, vocab_size=40, max_caption_len=6, device="cuda",
              transformer_dropout_prob=0.0, drop_prob=0.0, lr=1e-10, weight_decay=0.0)
    kw.update(over)
    opt = make_opt(**kw)
    model, criterion, _, _ = build(opt)
    model.load_state_dict(pdvc_state(f), strict=True)
    model = model.to(DEV)
    return opt, (model.train() if train else model.eval()), criterion, int(f["meta_T"])


def batches(T, seed=0):
    return [synth_batch(2, T, 64, 40, ns, DEV, seed=seed + i, cap_words=cw) for i, (ns, cw) in enumerate(LAYOUTS)]


def test_padded_criterion_equals_list_form():
    """criterion(outputs, PaddedTargets) == criterion(outputs, list of dicts): every loss, every matched index (one-to-one
    and the 4x-tiled many-to-one set), every gradient -- incl. videos with 0 and 1 events and unused slots."""
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    from gvl_amd.targets import PaddedTargets
    opt = make_opt(num_queries=20, feature_dim=64, vocab_size=40, device="cuda")
    _, criterion, _, _ = build(opt)
    g = torch.Generator().manual_seed(3)
    B, Q, nl = 4, 20, 2
    ns = [0, 3, 1, 6]

    def rnd(*s):
        return torch.randn(*s, generator=g).to(DEV)
    layers = [{"pred_logits": rnd(B, Q, 1), "pred_count": rnd(B, 11), "pred_boxes": torch.rand(B, Q, 2, generator=g).to(DEV) * 0.5 + 0.2}
              for _ in range(nl)]
    dt = synth_batch(B, 10, 64, 40, ns, DEV, seed=5)

    def run(targets):
        leaves = [{k: v.clone().requires_grad_() for k, v in o.items()} for o in layers]
        out = dict(leaves[0])
        out["aux_outputs"] = [dict(o) for o in leaves[1:]]
        losses, last, aux = criterion(out, targets)
        keys = sorted(k for k in losses if "self_iou" not in k)        # self-IoU is NaN for videos with < 2 events (0/0)
        total = sum(losses[k] * (i + 1) for i, k in enumerate(keys))
        total.backward()
        return losses, [last] + list(aux), leaves
    la, ma, ga = run(dt["video_target"])
    pt = PaddedTargets(B, 8, 0, DEV).load(dt)
    lb, mb, gb = run(pt)
    assert set(la) == set(lb)
    for k in la:
        a, b = float(la[k]), float(lb[k])
        assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-6 * max(1.0, abs(a)), k
    for x, y in zip(ma, mb):
        assert getattr(y.plan, "padded", False) and not getattr(x.plan, "padded", False)
        for kind in (0, 1):
            for (qa, ta), (qb, tb) in zip(x[kind], y[kind]):
                assert torch.equal(qa, qb) and torch.equal(ta, tb)
    for oa, ob in zip(ga, gb):
        for k in oa:
            assert maxerr(oa[k].grad, ob[k].grad) <= 1e-6 * max(1.0, float(oa[k].grad.abs().max())), k


@pytest.mark.parametrize("split", [False, True])
def test_graphed_train_step_one_graph_for_all_layouts(split):
    """Eight batches with different events per video (0..8) and caption widths: ONE captured graph replays them all and
    equals the eager step (losses, every gradient, Adam moments).  lr = 1e-10 keeps both sides' parameters identical
    (see test_graphed_train_step_equals_eager)."""
    from gvl_amd.parallel import TrainStep, GraphedTrainStep
    opt, model_a, crit_a, T = toy(True)
    _, model_b, crit_b, _ = toy(True)
    eager = TrainStep(model_a, crit_a, opt, capturable=True)
    graphed = GraphedTrainStep(model_b, crit_b, opt, split_exchange=split, max_gt=8, max_cap_len=8, cap_len_policy="grow")
    bs = batches(T)
    for step in range(12):
        dt = bs[step % len(bs)]
        la, loss_a = eager(dt)
        lb, loss_b = graphed(dt)
        assert abs(float(la) - float(lb)) < 1e-4 * max(1.0, abs(float(la))), (step, float(la), float(lb))
        for k in loss_a:
            a, b = float(loss_a[k]), float(loss_b[k])
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) < 1e-4 * max(1.0, abs(a)), (step, k, a, b)
        for (n, pa), pb in zip(model_a.named_parameters(), model_b.parameters()):
            if pa.grad is None:
                assert pb.grad is None or float(pb.grad.abs().max()) == 0.0, (step, n)
                continue
            assert maxerr(pa.grad, pb.grad) <= 1e-4 * max(1.0, float(pa.grad.abs().max())), (step, n)
    assert len(graphed.graphs) == 1 and graphed.captures == 1 and graphed.replays == 12
    sa, sb = eager.optimizer.state, graphed.optimizer.state
    for pa, pb in zip(eager.params, graphed.params):
        if pa.grad is None:
            continue
        # warm-up and capture left no trace: 12 batches -> 12 Adam steps on both sides
        assert float(sa[pa]["step"]) == float(sb[pb]["step"]) == 12.0
        assert maxerr(sa[pa]["exp_avg"], sb[pb]["exp_avg"]) <= 1e-4 * max(1.0, float(sa[pa]["exp_avg"].abs().max()))


def test_layout_keyed_capture_hides_no_parameter_from_the_update():
    """a layout-keyed (padded=False) capture is made from THIS rank's batch; a batch without events never
    reaches the captioner, and hiding "its" unused parameters from the captured clip + Adam would let the ranks of a
    data-parallel job apply different updates.  In the data-parallel (split) form such a capture therefore hides nothing:
    every parameter takes the step from the (all-reduced) flat gradient buffer, whatever the local batch looked like.  The
    padded capture -- static structure, the same on every rank -- may hide what its autograd graph never reaches."""
    from gvl_amd.parallel import GraphedTrainStep
    opt, model, crit, T = toy(True)
    bs = batches(T)
    g = GraphedTrainStep(model, crit, opt, split_exchange=True, padded=False, max_graphs=4)
    for dt in bs[:3]:                                       # incl. a batch with an event-less video
        g(dt)
        assert g.last_unused == [], [n for n, p in model.named_parameters() if any(p is u for u in g.last_unused)]
    # every trainable parameter was stepped by the captured update (Adam's step counter advanced for all of them)
    steps = {float(g.optimizer.state[p]["step"]) for p in g.params}
    assert len(steps) == 1 and steps.pop() == 3.0
    _, model2, crit2, _ = toy(True)
    g2 = GraphedTrainStep(model2, crit2, opt, split_exchange=True, max_gt=8, max_cap_len=8, cap_len_policy="grow")
    g2(bs[0])
    g2(bs[1])
    assert isinstance(g2.last_unused, list)                 # probed (possibly non-empty: rank-independent by construction)


def test_graphed_train_step_capacity_grows_and_cache_is_bounded():
    """Without a preset capacity the padded layout grows with the batches seen (each growth = one new capture, the
    superseded graph is dropped); the fallback form (padded=False: one graph per layout) is LRU-bounded."""
    from gvl_amd.parallel import GraphedTrainStep
    opt, model, crit, T = toy(True)
    bs = batches(T)
    g = GraphedTrainStep(model, crit, opt, cap_len_policy="grow")
    for dt in bs + bs:
        g(dt)
    assert len(g.graphs) == 1 and g.captures <= 4 and g.capacity.slots == 8
    # bucket policy: one graph per caption-width bucket of 4 tokens (widths 3..6 here -> buckets 4 and 8), each batch
    # replayed at its own bucket; growth of the grow-only capacities (slots, rows) drops the graphs it supersedes
    gb = GraphedTrainStep(model, crit, opt, cap_len_policy="bucket", cap_len_step=4)
    for dt in bs + bs:
        gb(dt)
    assert len(gb.graphs) == 2 and {k[3] for k in gb.graphs} == {4, 8} and gb.replays == 2 * len(bs)
    # the default: buckets of 2 tokens
    gd = GraphedTrainStep(model, crit, opt)
    for dt in bs + bs:
        gd(dt)
    assert gd.capacity.cap_len_policy == "bucket" and {k[3] for k in gd.graphs} <= {4, 6, 8} and len(gd.graphs) >= 2
    g2 = GraphedTrainStep(model, crit, opt, padded=False, max_graphs=3)
    for dt in bs:
        g2(dt)
    assert len(g2.graphs) == 3 and g2.captures == len(bs)


def test_warmup_and_capture_leave_training_state_untouched():
    """the first batch of a new graph must get ONE Adam update, like every other batch (train.py:405-409)."""
    from gvl_amd.parallel import GraphedTrainStep
    opt, model, crit, T = toy(True, lr=1e-3)
    before = [p.detach().clone() for p in model.parameters()]
    g = GraphedTrainStep(model, crit, opt, warmup=3)
    dt = batches(T)[0]
    snap = g._snapshot()
    g._capture(_loaded(dt))
    for p, b in zip(model.parameters(), before):
        assert torch.equal(p.detach(), b)
    for p in g.params:
        assert float(g.optimizer.state[p]["step"]) == 0.0 and float(g.optimizer.state[p]["exp_avg"].abs().max()) == 0.0
    assert torch.equal(torch.cuda.get_rng_state(torch.device(DEV)), snap[3])
    g(dt)
    assert all(float(g.optimizer.state[p]["step"]) == 1.0 for p in g.params if p.grad is not None)


def _loaded(dt):
    from gvl_amd.parallel import _PaddedBatch
    b = _PaddedBatch(dt, 8, 8)
    b.load(dt)
    return b.dt


@pytest.mark.parametrize("chunk", [0, 2])
def test_graphed_eval_forward_one_graph_for_all_layouts(chunk):
    """eval_utils.py:187-203: batches arrive with different numbers of events; one graph, results == eager forward."""
    from gvl_amd.parallel import GraphedEvalForward
    opt, model, crit, T = toy(False)
    ge = GraphedEvalForward(model, crit, max_gt=8, decode_chunk=chunk)
    for dt in batches(T, seed=20) + batches(T, seed=40)[:3]:
        with torch.no_grad():
            out_e, loss_e = model(dt, crit, None, "queries", eval_mode=True)
        out_g, loss_g = ge(dt)
        for k in ("pred_logits", "pred_boxes", "pred_count"):
            assert maxerr(out_g[k], out_e[k]) < 1e-5, k
        assert (len(out_g["seq"]) == 0 and len(out_e["seq"]) == 0) or torch.equal(out_g["seq"], out_e["seq"])
        if len(out_e["seq"]):
            assert maxerr(out_g["caption_probs"]["cap_prob_eval"], out_e["caption_probs"]["cap_prob_eval"]) < 1e-5
        for k in loss_e:
            a, b = float(loss_e[k]), float(loss_g[k])
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) < 1e-5 * max(1.0, abs(a)), k
        for kind in (0, 1):
            for (qa, ta), (qb, tb) in zip(out_e["matched_indices"][kind], out_g["matched_indices"][kind]):
                assert torch.equal(qa, qb) and torch.equal(ta, tb)
    assert len(ge.graphs) == 1 and ge.captures == 1


def test_decode_segments_stop_with_the_longest_caption():
    """LSTM_DSA.py:186-187: decoding ends when every sequence has ended.  With an <eos>-biased vocabulary layer the
    captions end after a few tokens: the segmented replay returns exactly the eager result and replays only the segments
    the longest caption needs."""
    from gvl_amd.parallel import GraphedEvalForward
    opt, model, crit, T = toy(False, max_caption_len=12)
    head = model.caption_head[-1]
    dt = batches(T, seed=60)[0]
    lengths = []
    for bias in (0.0, 1.5, 6.0):
        with torch.no_grad():
            head.logit.bias.zero_()
            head.logit.bias[0] = bias                        # token 0 = <eos>
            out_e, _ = model(dt, crit, None, "queries", eval_mode=True)
        ge = GraphedEvalForward(model, crit, decode_chunk=3)
        out_g, _ = ge(dt)
        full = GraphedEvalForward(model, crit, decode_chunk=0)
        out_f, _ = full(dt)
        n_e = out_e["seq"].shape[-1] if len(out_e["seq"]) else 0
        for o in (out_g, out_f):
            n = o["seq"].shape[-1] if len(o["seq"]) else 0
            assert n == n_e
            if n:
                assert torch.equal(o["seq"], out_e["seq"])
                assert maxerr(o["caption_probs"]["cap_prob_eval"], out_e["caption_probs"]["cap_prob_eval"]) < 1e-5
        # segments of 3 tokens: tokens [0,3) come with the main graph; one more segment per 3 further tokens (+1 token to
        # see the end), plus ONE speculative segment: the flags are read one segment behind the GPU
        needed = max(0, -(-(n_e + 1 - 3) // 3))
        expect = min(3, needed + 1)
        assert ge.segments_replayed == expect, (bias, n_e, ge.segments_replayed)
        lengths.append(n_e)
    assert lengths[0] >= lengths[1] >= lengths[2] and lengths[2] < 12


def test_graphed_eval_forward_without_criterion_or_targets():
    """pure inference (no ground truth in dt, criterion=None): one graph keyed on the tensor shapes alone"""
    from gvl_amd.parallel import GraphedEvalForward
    opt, model, crit, T = toy(False)
    ge = GraphedEvalForward(model, None)
    for i, dt in enumerate(batches(T, seed=80)[:3]):
        infer = {k: v for k, v in dt.items() if k in ("video_tensor", "video_mask", "video_length")}
        with torch.no_grad():
            out_e, _ = model(infer, None, None, "queries", eval_mode=True)
        out_g, loss_g = ge(infer)
        assert loss_g == {} and maxerr(out_g["pred_boxes"], out_e["pred_boxes"]) < 1e-5
        assert (len(out_g["seq"]) == 0 and len(out_e["seq"]) == 0) or torch.equal(out_g["seq"], out_e["seq"])
    assert len(ge.graphs) == 1 and ge.captures == 1


def test_graphed_eval_forward_beyond_the_padded_capacity_falls_back_and_returns():
    """a video with more events than queries (or more than 64) cannot use the padded layout
    (GraphedEvalForward._use_padded); such a batch gets a layout-keyed graph, results still equal the eager forward, and
    the next ordinary batch is served by the one padded graph again (no re-capture)."""
    from gvl_amd.parallel import GraphedEvalForward
    opt, model, crit, T = toy(False)                        # 8 queries
    ge = GraphedEvalForward(model, crit, max_gt=8, decode_chunk=0)    # (capacity preset: no growth captures in between)
    small = batches(T, seed=70)[:2]
    big = synth_batch(2, T, 64, 40, [11, 2], DEV, seed=75, cap_words=(3, 5))          # 11 events > 8 queries
    seen = []
    for dt in (small[0], big, small[1], big):
        with torch.no_grad():
            out_e, loss_e = model(dt, crit, None, "queries", eval_mode=True)
        out_g, loss_g = ge(dt)
        for k in ("pred_logits", "pred_boxes", "pred_count"):
            assert maxerr(out_g[k], out_e[k]) < 1e-5, k
        assert torch.equal(out_g["seq"], out_e["seq"])
        for k in loss_e:
            a, b = float(loss_e[k]), float(loss_g[k])
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) < 1e-5 * max(1.0, abs(a)), k
        seen.append((ge.captures, sorted(k[0] for k in ge.graphs)))
    # capture 1: padded graph; capture 2: the layout-keyed graph of the big batch; then both are replayed
    assert [c for c, _ in seen] == [1, 2, 2, 2], seen
    assert seen[-1][1] == ["layout", "padded"], seen


def test_caption_rows_kernel_equals_the_index_formulation(monkeypatch):
    """gvl_caption_rows (the captioner's pair rows on padded targets in one launch) against the PyTorch index formulation it
    replaces (GVL_CAPTION_ROWS=torch): every loss term and every gradient of the padded training forward, bit for bit -- over
    batches with 0 .. 8 events per video (unused rows, videos without events)."""
    opt, model, crit, T = toy(True)
    wd = crit.weight_dict
    for dt in batches(T, seed=70)[:4]:
        res = {}
        for mode in ("kernel", "torch"):
            monkeypatch.setenv("GVL_CAPTION_ROWS", "" if mode == "kernel" else "torch")
            model.zero_grad(set_to_none=True)
            out, loss = model(_loaded(dt), crit, None, "queries")
            final = sum(loss[k] * wd[k] for k in loss.keys() if k in wd)
            final.backward()
            res[mode] = ({k: v.detach().clone() for k, v in loss.items() if isinstance(v, torch.Tensor)},
                         {n: p_.grad.detach().clone() for n, p_ in model.named_parameters() if p_.grad is not None})
        (la, ga), (lb, gb) = res["kernel"], res["torch"]
        assert la.keys() == lb.keys() and ga.keys() == gb.keys()
        for k in la:
            assert torch.equal(la[k], lb[k]) or (torch.isnan(la[k]).all() and torch.isnan(lb[k]).all()), k
        for n in ga:
            # (the captioner's backward scatters with float atomics: equal to summation order)
            assert maxerr(ga[n], gb[n]) <= 1e-5 * max(1e-6, float(gb[n].abs().max())), n
