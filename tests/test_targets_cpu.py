"""Host logic of the layout-independent steps, on CPU: PaddedTargets (fixed-shape ground truth with per-video counts),
the grow-only / bucketed capacities and the bounded graph cache of gvl_amd.parallel.  (What reads these buffers on the
device is covered by tests/test_gpu_layout_independent.py.)"""
import pytest
import torch

from bench import synth_batch


def test_padded_targets_round_trip_and_refresh():
    from gvl_amd.targets import PaddedTargets, needed_capacity, total_events
    ns = [0, 3, 1, 5]
    dt = synth_batch(4, 6, 8, 30, ns, "cpu", seed=2, cap_words=(1, 4))
    assert needed_capacity(dt) == (5, dt["cap_tensor"].shape[1]) and total_events(dt) == 9
    pt = PaddedTargets(4, 8, 8, "cpu", pair_rows=32).load(dt)
    assert pt.counts.tolist() == ns and float(pt.num_boxes) == 9.0 and pt.host_counts == ns
    base = 0
    for v, n in enumerate(ns):
        assert torch.equal(pt.boxes[v, :n], dt["video_target"][v]["boxes"]) and float(pt.boxes[v, n:].abs().sum()) == 0
        w = dt["cap_tensor"].shape[1]
        assert torch.equal(pt.cap_tensor[v, :n, :w], dt["cap_tensor"][base:base + n])
        assert torch.equal(pt.cap_mask[v, :n, :w], dt["cap_mask"][base:base + n])
        assert int(pt.cap_tensor[v, n:].abs().sum()) == 0 and int(pt.cap_tensor[v, :, w:].abs().sum()) == 0
        base += n
    back = pt.as_list()
    assert [len(t_["boxes"]) for t_ in back] == ns and torch.equal(back[3]["boxes"], dt["video_target"][3]["boxes"])
    # refresh with another layout: slots of the previous batch must not survive
    dt2 = synth_batch(4, 6, 8, 30, [2, 0, 0, 1], "cpu", seed=3, cap_words=2)
    pt.load(dt2, num_boxes=2.5)
    assert pt.counts.tolist() == [2, 0, 0, 1] and float(pt.num_boxes) == 2.5
    assert float(pt.boxes[1].abs().sum()) == 0 and float(pt.boxes[3, 1:].abs().sum()) == 0
    assert int(pt.cap_tensor[3, 1:].abs().sum()) == 0
    # empty batch: the normaliser is floored at 1 (criterion.py:181)
    pt.load(synth_batch(4, 6, 8, 30, [0, 0, 0, 0], "cpu", seed=4))
    assert float(pt.num_boxes) == 1.0 and int(pt.counts.sum()) == 0


def test_padded_targets_reject_what_does_not_fit():
    from gvl_amd.targets import PaddedTargets
    dt = synth_batch(2, 6, 8, 30, [5, 1], "cpu", seed=2, cap_words=3)
    with pytest.raises(ValueError, match="cannot hold"):
        PaddedTargets(2, 4, 8, "cpu").load(dt)                       # 5 events > 4 slots
    with pytest.raises(ValueError, match="cap_tensor"):
        PaddedTargets(2, 8, 4, "cpu").load(dt)                       # caption tensor wider than cap_len
    with pytest.raises(ValueError, match="cannot hold"):
        PaddedTargets(2, 8, 8, "cpu", pair_rows=4).load(dt)          # 6 events > 4 caption rows
    assert not PaddedTargets(2, 4, 8, "cpu").fits(dt) and PaddedTargets(2, 8, 8, "cpu").fits(dt)


def test_capacity_policies_and_graph_cache():
    from gvl_amd.parallel import _Capacity, _LRU, _drop_superseded
    small = synth_batch(2, 6, 8, 30, [2, 3], "cpu", seed=1, cap_words=3)       # width 5
    wide = synth_batch(2, 6, 8, 30, [9, 1], "cpu", seed=1, cap_words=10)       # width 12
    grow = _Capacity()
    assert grow.fit(small, True) == (4, 8) and grow.pair_rows == 32
    assert grow.fit(wide, True) == (16, 12)
    assert grow.fit(small, True) == (16, 12)                                    # never shrinks
    assert _Capacity().fit(small, False) == (4, 0)                              # eval: no caption buffers
    bucket = _Capacity(cap_len_policy="bucket")
    assert bucket.fit(wide, True) == (16, 12) and bucket.fit(small, True) == (16, 8)    # width follows the batch, slots grow
    lru = _LRU(2)
    for k in "abc":
        lru.store(k, k.upper())
    assert list(lru) == ["b", "c"] and lru.lookup("b") == "B" and list(lru) == ["c", "b"] and lru.lookup("a") is None
    graphs = _LRU(8)
    sig = (("video_tensor", (2, 6, 8), "f32"),)
    for key in (("padded", sig, 4, 8, 32), ("padded", sig, 4, 12, 32), ("padded", "other", 4, 8, 32), ("layout", sig, 1)):
        graphs.store(key, object())
    _drop_superseded(graphs, ("padded", sig, 8, 8, 32), keep_width_buckets=True)      # slots grew: both widths are dead
    assert set(graphs) == {("padded", "other", 4, 8, 32), ("layout", sig, 1)}
    graphs.store(("padded", sig, 8, 8, 32), object())
    _drop_superseded(graphs, ("padded", sig, 8, 12, 32), keep_width_buckets=True)     # only the width differs: stays
    assert ("padded", sig, 8, 8, 32) in graphs
    _drop_superseded(graphs, ("padded", sig, 8, 12, 32), keep_width_buckets=False)    # grow policy: it is superseded
    assert ("padded", sig, 8, 8, 32) not in graphs


def test_graph_replay_guard_detects_a_late_environment_setting():
    """DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 must be in place before the HIP runtime initialises; gvl_amd records when it
    was not and the captured steps refuse to capture (a fresh interpreter per case: the flag is computed at import)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "import gvl_amd\n"
            "print(int(gvl_amd.GRAPH_REPLAY_UNSAFE), os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'))\n"
            "try:\n    gvl_amd.graph_replay_guard('x'); print('ok')\nexcept RuntimeError as e:\n    print('raised')\n") % root

    def run(env_value, extra=None):
        env = dict(os.environ)
        env.pop("DEBUG_CLR_GRAPH_PACKET_CAPTURE", None)
        env.pop("GVL_ALLOW_GRAPH_PACKET_CAPTURE", None)
        if env_value is not None:
            env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = env_value
        env.update(extra or {})
        return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300).stdout.split()
    assert run(None) == ["0", "0", "ok"]                      # unset, runtime not up: gvl_amd sets it in time
    assert run("0") == ["0", "0", "ok"]
    assert run("1") == ["1", "1", "raised"]                   # an explicit non-zero setting is respected -- and refused for captures
    assert run("1", {"GVL_ALLOW_GRAPH_PACKET_CAPTURE": "1"}) == ["1", "1", "ok"]
