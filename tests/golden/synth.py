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
