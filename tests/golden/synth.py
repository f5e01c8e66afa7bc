"""Deterministic synthetic weights / inputs shared by the golden-vector generator and the tests.

Weights of the d_model=512 transformer are far too large to commit, so the fixtures store only inputs-by-seed
and expected outputs; both sides regenerate identical weights from the parameter *name* with numpy's frozen
legacy ``RandomState`` stream (bit-stable across numpy versions and machines).
"""
import zlib

import numpy as np


def _rs(key, seed):
    return np.random.RandomState((zlib.crc32(key.encode()) + 7919 * seed) % (2 ** 32))


def synth_tensor(key, shape, seed=0):
    """float32 array for parameter `key` of `shape`."""
    rs = _rs(key, seed)
    shape = tuple(int(s) for s in shape)
    if len(shape) >= 2:
        if key.endswith("embed.weight") or key.endswith("level_embed"):
            return rs.standard_normal(shape).astype(np.float32)
        fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
        a = np.sqrt(6.0 / (fan_in + fan_out))
        return rs.uniform(-a, a, shape).astype(np.float32)
    if key.endswith("sampling_offsets.bias"):
        return rs.uniform(-2.0, 2.0, shape).astype(np.float32)
    if key.endswith(".weight"):                        # LayerNorm / GroupNorm scale
        return (1.0 + 0.1 * rs.uniform(-1, 1, shape)).astype(np.float32)
    return (0.05 * rs.uniform(-1, 1, shape)).astype(np.float32)


def synth_state_dict(shapes, seed=0):
    """shapes: {name: shape} -> {name: float32 ndarray}"""
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items()}


def synth_array(tag, shape, seed=0, lo=None, hi=None):
    """Named input array: N(0,1) by default, U(lo,hi) when bounds are given."""
    rs = _rs("input:" + tag, seed)
    if lo is None:
        return rs.standard_normal(tuple(shape)).astype(np.float32)
    return rs.uniform(lo, hi, tuple(shape)).astype(np.float32)


def level_lengths(T, n_levels=4):
    """T_l of the stride-2 conv pyramid: floor((T-1)/2)+1 per level (base_encoder.py:39)."""
    out = [T]
    for _ in range(n_levels - 1):
        out.append((out[-1] - 1) // 2 + 1)
    return out


def synth_samples(seed=3):
    """per-video samples in the layout of the reference dataset's __getitem__ (video_dataset.py:collate_fn input)"""
    rs = np.random.RandomState(seed)
    out = []
    for i, (T, n) in enumerate([(37, 3), (52, 1), (20, 4)]):
        feats = rs.standard_normal((T, 16)).astype(np.float32)
        dur = 40.0 + 13.5 * i
        starts = np.sort(rs.uniform(0, dur * 0.7, n))
        raw_ts = [[float(s_), float(min(dur, s_ + rs.uniform(2.0, dur * 0.3)))] for s_ in starts]
        featstamps = [[int(a / dur * T), int(b / dur * T)] for a, b in raw_ts]
        caps = [rs.randint(1, 50, size=rs.randint(3, 9)).astype(np.int64) for _ in range(n)]
        for c in caps:
            c[0] = 0
            c[-1] = 0
        raw = ["caption %d %d" % (i, k) for k in range(n)]
        out.append((feats, featstamps, [0] * n, caps, raw_ts, dur, raw, "v_%03d" % i))
    return out


def synth_dataset(root, seed=5):
    """A tiny on-disk dataset in the reference's formats (annotation JSON, vocabulary JSON, one .npy per video) for the
    PropSeqDataset fixtures: videos whose feature length equals / exceeds / is below the target length, a single-frame
    video, a video WITHOUT a feature file, and one with more events than gt_proposal_sample_num.
    -> dict(anno, vocab, tsp_dir, c3d_dir, vocab_size)"""
    import json
    import os
    rs = np.random.RandomState(seed)
    words = ["a", "man", "woman", "is", "are", "the", "dog", "runs", "jumps", "then", "she", "he", "cooks", "plays",
             "guitar", "outside", "inside", "while", "people", "watch", ".", ","]
    vocab = {"word_to_ix": {w: i + 1 for i, w in enumerate(words)}, "ix_to_word": {str(i + 1): w for i, w in enumerate(words)}}
    os.makedirs(os.path.join(root, "tsp"), exist_ok=True)
    os.makedirs(os.path.join(root, "c3d"), exist_ok=True)
    anno = {}
    for i, (T, n) in enumerate([(20, 2), (33, 5), (1, 1), (None, 3), (11, 4)]):
        key = "v_%011d" % (i * 7 + 3)                       # 13 characters, as ActivityNet ids
        dur = 35.0 + 11.25 * i
        starts = np.sort(rs.uniform(0, dur * 0.8, n))
        stamps = [[round(float(s_), 2), round(float(min(dur + 3.0, s_ + rs.uniform(1.0, dur * 0.4))), 2)] for s_ in starts]
        sents = [" ".join(rs.choice(words[:20] + ["unknownword"], size=rs.randint(3, 12))).capitalize() + "." for _ in range(n)]
        anno[key] = {"duration": dur, "timestamps": stamps, "sentences": sents}
        if T is not None:
            np.save(os.path.join(root, "tsp", key + ".npy"), rs.standard_normal((T, 512)).astype(np.float32))
            np.save(os.path.join(root, "c3d", key + ".npy"), rs.standard_normal((T, 500)).astype(np.float32))
    with open(os.path.join(root, "anno.json"), "w") as f:
        json.dump(anno, f)
    with open(os.path.join(root, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    return {"anno": os.path.join(root, "anno.json"), "vocab": os.path.join(root, "vocab.json"),
            "tsp_dir": os.path.join(root, "tsp"), "c3d_dir": os.path.join(root, "c3d"), "vocab_size": len(words)}


def dataset_opt(kind, vocab_size):
    """options PropSeqDataset reads, for the two feature-type branches of load_feats (list-typed 'tsp' as
    cfgs/anet_tsp_ssvg.yml, scalar 'c3d' with data_norm as cfgs/anet_c3d_ssvg.yml allows)"""
    import argparse
    common = dict(vocab_size=vocab_size, max_caption_len=8, invalid_video_json=[], feature_sample_rate=1,
                  train_proposal_sample_num=24, gt_proposal_sample_num=3, num_queries=10, data_rescale=1,
                  frame_embedding_num=20, data_norm=0)
    if kind == "tsp":
        common.update(visual_feature_type=["tsp"], feature_dim=512)
    else:
        common.update(visual_feature_type="c3d", feature_dim=500, data_norm=1)
    return argparse.Namespace(**common)
