"""End-to-end evaluation flow on gvl_amd alone, the way eval_utils.py:180-239 drives the reference:
samples -> collate_fn -> device -> PDVC.forward (eager and as one hipGraph) -> PostProcess -> result-file records."""
import json

import pytest
import torch

from synth import synth_samples

pytestmark = pytest.mark.gpu


class _Translator:
    @staticmethod
    def rtranslate(s):
        return " ".join(str(int(x)) for x in s if x > 0)


LOADER = type("L", (), {"dataset": type("D", (), {"translator": _Translator})})


def to_device(dt, dev):
    out = dict(dt)
    for k, v in dt.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.to(dev)
    out["video_target"] = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in tg.items()}
                           for tg in dt["video_target"]]
    return out


def test_samples_to_result_file(tmp_path):
    from gvl_amd.config import make_opt
    from gvl_amd.eval_utils import batch_result_json, new_result_file, save_dvc_json
    from gvl_amd.parallel import GraphedEvalForward
    from gvl_amd.pdvc import build
    from gvl_amd.video_dataset import collate_fn
    dev = torch.device("cuda:0")
    opt = make_opt(num_queries=10, feature_dim=16, vocab_size=60, max_caption_len=7, frame_embedding_num=64,
                   device="cuda")
    torch.manual_seed(0)
    model, criterion, _, post = build(opt)
    model = model.to(dev).eval()
    dt = to_device(collate_fn(synth_samples()), dev)
    files = []
    graphed = GraphedEvalForward(model, criterion)
    for mode in ("eager", "graph", "graph"):
        with torch.no_grad():
            out, loss = model(dt, criterion, None, "queries", eval_mode=True) if mode == "eager" else graphed(dt)
        assert all(torch.isfinite(v).all() or k.startswith("loss_self_iou") for k, v in loss.items())   # 0/0 for the single-event video, as the reference
        results = post["bbox"](out, dt["video_length"][:, 1], LOADER)
        rec = new_result_file()
        rec["results"].update(batch_result_json(results, dt["video_key"], score_threshold=0))
        path = tmp_path / f"{mode}_{len(files)}.json"
        save_dvc_json(rec, str(path), verbose=True)
        files.append(json.load(open(path)))
    ref = files[0]
    assert sorted(ref["results"]) == ["v_000", "v_001", "v_002"] and ref["valid_video_num"] == 3
    for vid, events in ref["results"].items():
        assert len(events) == 10
        for e in events:
            assert 0.0 <= e["timestamp"][0] <= e["timestamp"][1] <= e["vid_duration"] + 1e-4
            assert isinstance(e["sentence"], str) and 1 <= e["pred_event_count"] <= 10
    for other in files[1:]:                                   # graph replays reproduce the eager records
        for vid in ref["results"]:
            for a, b in zip(ref["results"][vid], other["results"][vid]):
                assert a["sentence"] == b["sentence"] and a["query_id"] == b["query_id"]
                assert abs(a["proposal_score"] - b["proposal_score"]) < 1e-5
                assert max(abs(x - y) for x, y in zip(a["timestamp"], b["timestamp"])) < 1e-3


@pytest.mark.parametrize("B,T,Q,n_gt", [(1, 37, 30, [2]), (3, 64, 7, [1, 0, 4]), (2, 129, 33, [5, 3]), (5, 20, 12, [1, 1, 2, 3, 1])])
def test_shape_sweep_eval_and_train(B, T, Q, n_gt):
    """odd shapes through the whole model: batch 1, query counts that are not multiples of 4, ragged video lengths,
    a video without events, T where the level lengths are odd -- eager == graphed eval, and one captured train step
    with finite losses and gradients."""
    import numpy as np
    from gvl_amd.config import make_opt
    from gvl_amd.parallel import GraphedEvalForward, GraphedTrainStep
    from gvl_amd.pdvc import build
    dev = torch.device("cuda:0")
    opt = make_opt(num_queries=Q, feature_dim=24, vocab_size=50, max_caption_len=6, frame_embedding_num=T, device="cuda")
    torch.manual_seed(B * 100 + T)
    model, criterion, _, _ = build(opt)
    model = model.to(dev)
    g = torch.Generator().manual_seed(T)
    valid = [T] + [int(x) for x in torch.randint(max(2, T // 3), T + 1, (B - 1,), generator=g)]
    vt = torch.randn(B, T, 24, generator=g)
    vmask = torch.zeros(B, T, dtype=torch.bool)
    for i, v in enumerate(valid):
        vmask[i, :v] = True
        vt[i, v:] = 0
    targets = []
    for n in n_gt:
        c = torch.rand(n, generator=g) * 0.5 + 0.25
        l_ = torch.rand(n, generator=g) * 0.3 + 0.1
        targets.append({"boxes": torch.stack([c, l_], -1).to(dev), "labels": torch.zeros(n, dtype=torch.long, device=dev)})
    ncap = sum(n_gt)
    caps = torch.randint(1, 50, (ncap, 6), generator=g)
    caps[:, 0] = 0
    caps[:, -1] = 0
    dt = {"video_tensor": vt.to(dev), "video_mask": vmask.to(dev),
          "video_length": torch.tensor([[float(v), 50.0 + i, float(n)] for i, (v, n) in enumerate(zip(valid, n_gt))]).to(dev),
          "video_target": targets, "cap_raw": [["x"] * n for n in n_gt], "cap_tensor": caps.to(dev),
          "cap_mask": (caps != 0).float().to(dev).index_fill_(1, torch.tensor([0], device=dev), 1.0),
          "gt_boxes_mask": torch.ones(B, max(1, max(n_gt)), dtype=torch.bool, device=dev)}
    model.eval()
    with torch.no_grad():
        ref_out, ref_loss = model(dt, criterion, None, "queries", eval_mode=True)
    out, loss = GraphedEvalForward(model, criterion)(dt)
    for k in ("pred_logits", "pred_boxes", "pred_count"):
        assert torch.isfinite(ref_out[k]).all() and float((out[k] - ref_out[k]).abs().max()) < 1e-5, k
    assert torch.equal(out["seq"], ref_out["seq"])
    model.train()
    step = GraphedTrainStep(model, criterion, opt, warmup=1)
    for _ in range(2):
        final, losses = step(dt)
    assert bool(torch.isfinite(final))
    bad = [k for k, v in losses.items() if not torch.isfinite(v).all() and not k.startswith("loss_self_iou")]
    assert not bad, bad
    assert all(torch.isfinite(p_.grad).all() for p_ in model.parameters() if p_.grad is not None)
