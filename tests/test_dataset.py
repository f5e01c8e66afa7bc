"""SURVEY.md section 8 row f4, dataset side: gvl_amd.video_dataset.PropSeqDataset against the reference's
PropSeqDataset.__getitem__ (video_dataset.py:209-281, nearest rescale :386-397, zeros for missing files :319-322) on
the synthetic on-disk dataset that tests/golden/make_golden.py:make_dataset fed to the imported reference."""
import numpy as np
import torch

from helpers import load
from synth import synth_dataset, dataset_opt


def test_prop_seq_dataset_matches_reference(tmp_path):
    from gvl_amd.video_dataset import PropSeqDataset, Translator, collate_fn
    g = load("dataset")
    info = synth_dataset(str(tmp_path))
    for kind in ("tsp", "c3d"):
        opt = dataset_opt(kind, info["vocab_size"])
        folder = [info["tsp_dir"]] if kind == "tsp" else info["c3d_dir"]
        ds = PropSeqDataset(info["anno"], folder, info["vocab"], True, "gt", opt)
        assert len(ds) == int(g[f"{kind}.n"])
        np.random.seed(7)                                      # the event sub-sampling uses numpy's global RNG (:268)
        samples = [ds[i] for i in range(len(ds))]
        for i, (feats, featstamps, labels, caps, stamps, dur, raw, key) in enumerate(samples):
            pre = f"{kind}.{i}."
            want = g[pre + "feats"]
            assert feats.shape == want.shape == (opt.frame_embedding_num, opt.feature_dim)
            assert np.array_equal(np.asarray(feats, dtype=want.dtype), want), (kind, i)      # a gather: bit-exact
            assert np.array_equal(np.asarray(featstamps).reshape(-1, 2), g[pre + "featstamps"])
            assert list(labels) == g[pre + "labels"].tolist()
            assert [len(c) for c in caps] == g[pre + "cap_lens"].tolist()
            assert np.array_equal(np.concatenate(caps), g[pre + "caps"])
            assert np.array_equal(np.asarray(stamps, dtype=np.float64).reshape(-1, 2), g[pre + "stamps"])
            assert float(dur) == float(g[pre + "duration"]) and key == str(g[pre + "key"])
            assert list(raw) == [str(x) for x in g[pre + "raw"]]
        dt = collate_fn(samples[:3])
        assert torch.equal(dt["video_tensor"], torch.from_numpy(g[f"{kind}.collate.video_tensor"]))
        assert torch.equal(dt["cap_tensor"], torch.from_numpy(g[f"{kind}.collate.cap_tensor"]))
        assert torch.equal(dt["video_length"], torch.from_numpy(g[f"{kind}.collate.video_length"]))
    tr = Translator(info["vocab"], info["vocab_size"])
    assert [tr.rtranslate([3, 7, 0, 5]), tr.rtranslate([0, 1]), tr.rtranslate([2, 22, 4])] == [str(x) for x in g["rtranslate"]]


def test_resize_feature_rounds_half_down_like_interp1d():
    """scipy interp1d(kind='nearest') rounds x.5 DOWN; checked against scipy itself on lengths that hit exact halves"""
    from scipy.interpolate import interp1d
    from gvl_amd.video_dataset import resize_feature
    rs = np.random.RandomState(0)
    for T, new in [(3, 5), (5, 9), (2, 3), (11, 21), (33, 20), (100, 20), (7, 100), (1, 4)]:
        x = rs.standard_normal((T, 6)).astype(np.float32)
        if T == 1:
            want = np.stack([x.reshape(-1)] * new)
        else:
            want = interp1d(np.arange(T), x, axis=0, kind="nearest")([i * float(T - 1) / (new - 1) for i in range(new)])
        assert np.array_equal(resize_feature(x, new), want.astype(np.float32)), (T, new)
