"""Operand planes of the training step's weights (include/gvl_msda.h: gvl_planes_refresh_f16).

The hand-written Linear products of the TRAINING step (gvl_amd/linear.py: forward ``x W^T + b`` and input gradient ``dy W`` on
``gvl_linear_f16x3_f32``) read their weight operand as fp16 (hi, 2^11 lo) planes -- of ``W`` for the forward, of ``W^T`` for the
input gradient.  The weights change with every optimizer step (train.py:405-409), so the planes are rebuilt once per training
forward: ``TrainPlanes.refresh()`` does that for ALL registered matrices with two launches (a per-group maximum, then split +
transpose), ``PDVC.forward`` calls it when the model trains.  A weight that is not registered (a module used on its own) gets its
planes lazily, per parameter version, by ``gvl_split_rows_f16`` -- same numbers, two launches per matrix and orientation.

An *operand* is one or more (N_i, K) matrices stacked along N (``[sampling_offsets ; attention_weights]``, the q / k / v blocks of
``nn.MultiheadAttention.in_proj_weight`` are one matrix already) that one GEMM launch multiplies; all blocks of an operand share
one power-of-two scale (the transposed planes contract over the stacked rows)."""
import ctypes

import numpy as np
import torch

from . import _lib
from . import MultiScaleDeformableAttention as MSDA


class _Desc(ctypes.Structure):                      # include/gvl_msda.h: gvl_plane_desc
    _fields_ = [("w", ctypes.c_void_p), ("N", ctypes.c_int), ("K", ctypes.c_int), ("hi", ctypes.c_void_p), ("lo", ctypes.c_void_p),
                ("scale", ctypes.c_void_p), ("n_total", ctypes.c_int), ("n_off", ctypes.c_int), ("t_hi", ctypes.c_void_p),
                ("t_lo", ctypes.c_void_p), ("t_scale", ctypes.c_void_p), ("group_chunk_begin", ctypes.c_int),
                ("group_chunks", ctypes.c_int), ("bias", ctypes.c_void_p), ("bias_dst", ctypes.c_void_p)]


class Operand:
    """what gvl_amd.layers.linear takes as `w`: planes (hi, lo, scale) of an (N, K) operand + its bias (or None)"""

    def __init__(self, planes, N, K, bias=None):
        self.planes, self.N, self.K, self.bias = planes, N, K, bias


def eligible_matrix(w, any_rows=False):
    return (isinstance(w, torch.Tensor) and w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.is_contiguous()
            and (any_rows or w.shape[0] % 32 == 0) and w.shape[1] % 32 == 0 and w.data_ptr() % 16 == 0)


class TrainPlanes:
    """planes of registered operands, both orientations; ``refresh()`` = two launches for all of them"""

    def __init__(self, device):
        self.device = torch.device(device)
        self.operands = []                 # [(key, [weights], fwd Operand planes, transposed Operand planes)]
        self.by_key = {}
        self._built = None
        self.mirrors = []                  # [(contiguous buffer, callable -> the strided piece of a parameter it mirrors)]: see register_mirror

    @staticmethod
    def key_of(weights):
        return tuple(id(w) for w in weights)

    def register(self, weights, biases=None):
        """weights: list of (N_i, K) parameters stacked along N (biases: the matching (N_i) parameters or None each)
        -> index of the operand"""
        weights = list(weights)
        biases = list(biases) if biases is not None else [None] * len(weights)
        key = self.key_of(weights)
        if key in self.by_key:
            return self.by_key[key]
        if not all(eligible_matrix(w, any_rows=(i == len(weights) - 1)) for i, w in enumerate(weights)) \
                or len({w.shape[1] for w in weights}) != 1:
            raise ValueError("TrainPlanes.register: fp32 CUDA matrices (N_i, K) with K and every N_i but the last multiples of 32")
        K = weights[0].shape[1]
        n_total = sum(w.shape[0] for w in weights)
        fwd = MSDA.SplitPlanes(n_total, K, self.device)
        tr = MSDA.SplitPlanes(K, (n_total + 31) // 32 * 32, self.device)       # (contraction padded with zeros to a K stage)
        bias = torch.zeros(n_total, device=self.device, dtype=torch.float32) if any(b is not None for b in biases) else None
        self.operands.append((key, weights, fwd, tr, biases, bias))
        self.by_key[key] = len(self.operands) - 1
        self._built = None
        return self.by_key[key]

    def register_mirror(self, buf, source):
        """a COLUMN BLOCK of a parameter as an operand: the planes' descriptors address contiguous matrices only, so the block is
        copied into `buf` (contiguous, same shape) at the top of every refresh() and `buf` is what is registered and looked up;
        source(): the current strided view (e.g. the embedding columns of the LSTM's W_ih: LSTM_DSA.py:267-269)"""
        self.register([buf], [None])
        self.mirrors.append((buf, source))

    def lookup(self, weights):
        i = self.by_key.get(self.key_of(weights))
        if i is None:
            return None
        _, ws, fwd, tr, _, bias = self.operands[i]
        if any(a is not b for a, b in zip(ws, weights)):          # (an id() re-used by another tensor)
            return None
        return fwd, tr, bias

    def _build(self):
        L = _lib.lib()
        chunk = L.gvl_planes_chunk_elems()
        descs, chunk_map, wg_map = [], [], []
        for _, ws, fwd, tr, bs, bias in self.operands:
            group_begin = len(chunk_map)
            first = len(descs)
            n_off = 0
            n_total = sum(w.shape[0] for w in ws)
            for w, b in zip(ws, bs):
                N, K = w.shape
                di = len(descs)
                descs.append(_Desc(w.data_ptr(), N, K, fwd.hi.data_ptr(), fwd.lo.data_ptr(), fwd.scale.data_ptr(), n_total, n_off,
                                   tr.hi.data_ptr(), tr.lo.data_ptr(), tr.scale.data_ptr(), 0, 0,
                                   b.data_ptr() if b is not None else None, bias.data_ptr() if bias is not None else None))
                for c in range((N * K + chunk - 1) // chunk):
                    chunk_map.append((di, c))
                tiles = ((N + 31) // 32) * (K // 32)
                for t in range(0, tiles, 4):
                    wg_map.append((di, t))
                n_off += N
            for d in descs[first:]:
                d.group_chunk_begin, d.group_chunks = group_begin, len(chunk_map) - group_begin
        arr = (_Desc * len(descs))(*descs)
        raw = np.frombuffer(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr)), dtype=np.uint8).copy()
        self._descs = torch.from_numpy(raw).to(self.device)
        self._chunk_map = torch.tensor(chunk_map, dtype=torch.int32, device=self.device).contiguous()
        self._wg_map = torch.tensor(wg_map, dtype=torch.int32, device=self.device).contiguous()
        self._chunk_amax = torch.zeros(len(chunk_map), dtype=torch.float32, device=self.device)
        self._built = (self._ptrs(), len(chunk_map), len(wg_map))

    def refresh(self):
        """rebuild the planes of every registered operand from the parameters' current values (graph-capturable)"""
        if not self.operands:
            return
        with torch.no_grad():
            for buf, source in self.mirrors:
                buf.copy_(source())
        if self._built is None or self._built[0] != self._ptrs():        # (a parameter was re-allocated: .to(), load with assign)
            self._build()
        with torch.cuda.device(self.device):
            rc = _lib.lib().gvl_planes_refresh_f16(self._descs.data_ptr(), self._chunk_map.data_ptr(), self._built[1],
                                                   self._wg_map.data_ptr(), self._built[2], self._chunk_amax.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "planes_refresh")
        self.fresh_for = self._versions()

    def _ptrs(self):
        return tuple(t.data_ptr() for op in self.operands for t in list(op[1]) + [b for b in op[4] if b is not None])

    def _versions(self):
        return tuple(t._version for op in self.operands for t in list(op[1]) + [b for b in op[4] if b is not None])

    def is_fresh(self):
        return getattr(self, "fresh_for", None) == self._versions()


def build_train_planes(model, device):
    """register every operand of `model` that gvl_amd.linear.train_linear multiplies: the gvl_amd Linear layers, the stacked
    [sampling_offsets ; attention_weights] projection of every MSDeformAttn, the in / out projections of nn.MultiheadAttention"""
    from .linear import Linear
    from .ops.modules import MSDeformAttn
    tp = TrainPlanes(device)
    seen = set()

    def add(ws, bs):
        if all(eligible_matrix(w) and w.device == tp.device for w in ws) and sum(w.shape[0] for w in ws) % 64 == 0 \
                and ws[0].shape[1] % 64 == 0 and TrainPlanes.key_of(ws) not in seen:
            seen.add(TrainPlanes.key_of(ws))
            tp.register(ws, bs)
    for m in model.modules():
        if isinstance(m, Linear):
            if m.weight.shape[0] % 32 and eligible_matrix(m.weight, any_rows=True) and TrainPlanes.key_of([m.weight]) not in seen:
                # the vocabulary layer (8518 rows): its products are gvl_gemm_f16x3_f32 (forward) and the split-K input gradient
                seen.add(TrainPlanes.key_of([m.weight]))
                tp.register([m.weight], [m.bias])
                continue
            add([m.weight], [m.bias])
        elif isinstance(m, MSDeformAttn):
            add([m.sampling_offsets.weight, m.attention_weights.weight], [m.sampling_offsets.bias, m.attention_weights.bias])
        elif isinstance(m, torch.nn.MultiheadAttention) and m.in_proj_weight is not None:
            add([m.in_proj_weight], [m.in_proj_bias])
            add([m.out_proj.weight], [m.out_proj.bias])
        elif type(m).__name__ == "ShowAttendTellCore" and hasattr(m, "rnn"):
            # the embedding columns of W_ih multiply every teacher-forced token's embedding at once (4416 x 512 x 2048 at cfg A: 100 us
            # forward + 115 us backward on the fp32 library GEMMs): a mirror of that column block is an operand of its own
            w, E = m.rnn.weight_ih_l0, m.input_encoding_size
            if w.device == tp.device and w.dtype == torch.float32 and w.shape[0] % 64 == 0 and E % 64 == 0 and E < w.shape[1]:
                buf = torch.empty(w.shape[0], E, device=tp.device, dtype=torch.float32)
                m.__dict__["_gvl_wx_mirror"] = buf
                tp.register_mirror(buf, lambda m=m, E=E: m.rnn.weight_ih_l0[:, :E])
    return tp
