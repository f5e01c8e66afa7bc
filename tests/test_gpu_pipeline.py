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
