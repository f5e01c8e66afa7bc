"""The update of the training step -- ``clip_grad_norm_(params, max_norm); optimizer.step()`` of train.py:405-409 with a
``torch.optim.Adam`` -- on gvl_clip_adam_step_f32 (gvl_optim.hip): three launches over a table of (parameter, gradient, moments)
instead of torch's ~10 multi-tensor launches that pass over the gradients three times.  The optimizer object, its hyper-parameters
and its STATE stay torch's (``state_dict`` / checkpoints unchanged): this only replaces what one clip + step computes.
``GVL_OPTIM=torch`` keeps torch's own launches (A/B switch); anything this path does not cover (several parameter groups, amsgrad,
maximize, non-fp32 or non-contiguous tensors, state not yet created) falls back to them as well -- same numbers either way, to fp32
rounding (tests/test_gpu_optim.py)."""
import ctypes
import os

import numpy as np
import torch

from . import _lib

_ENABLED = os.environ.get("GVL_OPTIM", "") != "torch"


class _Desc(ctypes.Structure):                      # include/gvl_msda.h: gvl_adam_desc
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("step", ctypes.c_void_p), ("n", ctypes.c_int64), ("vec", ctypes.c_int), ("pad_", ctypes.c_int)]


def bump_versions(tensors):
    """The kernels of this file -- and every hipGraph replay of a captured update -- write parameters through raw pointers, so
    torch's version counters do not move by themselves.  Every weight-derived cache of the package (split-fp16 planes, the
    captioner's stacked matrices, captured eval / decode graphs, reference-point tables) is keyed on (data_ptr, _version): without
    this bump an evaluation after training would run on the planes of the weights it saw first.  No launch, no
    synchronisation: the counter is host-side metadata, as torch's own in-place update would have moved it."""
    torch.autograd.graph.increment_version(tensors)


class ClipAdam:
    def __init__(self, optimizer, max_norm):
        # max_norm reaches clip_grad_norm_'s formula unchanged (0 zeroes the gradients, as torch's own call with 0 does)
        self.opt, self.max_norm = optimizer, float(max_norm)
        self.tables = {}
        self.last = None                                     # scal tensor of the last step: [total norm, clip coefficient, ...]

    def _group(self):
        if not isinstance(self.opt, torch.optim.Adam) or isinstance(self.opt, torch.optim.AdamW) or len(self.opt.param_groups) != 1:
            return None
        g = self.opt.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or isinstance(g["lr"], torch.Tensor):
            return None
        return g

    def _state(self, params):
        """[(p, grad, exp_avg, exp_avg_sq, step)] or None when something is not what the kernels take"""
        rows = []
        for p in params:
            st = self.opt.state.get(p)
            if not st or "exp_avg" not in st:
                return None
            g, m, v, t = p.grad, st["exp_avg"], st["exp_avg_sq"], st["step"]
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
                return None
            for x in (p, g, m, v):
                if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.device == p.device
                        and x.numel() == p.numel() and x.data_ptr() % 4 == 0):
                    return None
            rows.append((p, g, m, v, t))
        return rows

    def _build(self, rows, vec_grads):
        chunk = _lib.lib().gvl_adam_chunk_elems()
        dev = rows[0][0].device
        descs, cmap = [], []
        for i, (p, g, m, v, _) in enumerate(rows):
            n = p.numel()
            vec = int(vec_grads and n % 4 == 0 and all(x.data_ptr() % 16 == 0 for x in (p, m, v)))
            descs.append(_Desc(p.data_ptr(), None, m.data_ptr(), v.data_ptr(), rows[i][4].data_ptr(), n, vec, 0))
            cmap.extend((i, c) for c in range((n + chunk - 1) // chunk))
        arr = (_Desc * len(descs))(*descs)
        raw = np.frombuffer(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr)), dtype=np.uint8).copy()
        return {"descs": torch.from_numpy(raw).to(dev), "cmap": torch.tensor(cmap, dtype=torch.int32, device=dev).contiguous(),
                "partial": torch.empty(len(cmap), dtype=torch.float32, device=dev),
                "scal": torch.zeros(4, dtype=torch.float32, device=dev),
                "corr": torch.zeros(2 * len(rows), dtype=torch.float32, device=dev), "n_chunks": len(cmap),
                "ptrs": tuple(x.data_ptr() for r in rows for x in (r[0], r[2], r[3], r[4])), "steps": [r[4] for r in rows],
                "n": len(rows)}

    def step(self, params):
        """clip + Adam over `params` (those that have a gradient) -> True, or False when torch's own path must run"""
        g = self._group()
        if not _ENABLED or g is None or not params or not params[0].is_cuda:
            return False
        rows = self._state(params)
        if rows is None:
            return False
        # gradients 16-byte aligned (fresh tensors: always; views of a flat buffer: not all): the table's `vec` flags count on it
        vec_grads = all(r[1].data_ptr() % 16 == 0 for r in rows)
        key = (tuple(id(p) for p in params), vec_grads)
        tab = self.tables.get(key)
        ptrs = tuple(x.data_ptr() for r in rows for x in (r[0], r[2], r[3], r[4]))
        if tab is None or tab["ptrs"] != ptrs:
            if torch.cuda.is_current_stream_capturing():         # (the table is built with host -> device copies)
                return False
            tab = self.tables[key] = self._build(rows, vec_grads)
        # the gradients' addresses change (a step without a flat buffer gets fresh tensors from autograd; a captured step's live
        # in the graph's pool): they reach the table through the arguments of a launch
        gp = (ctypes.c_void_p * tab["n"])(*[r[1].data_ptr() for r in rows])
        with torch.cuda.device(params[0].device):
            rc = _lib.lib().gvl_adam_set_grads(tab["descs"].data_ptr(), tab["n"], gp, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "adam_set_grads")
        torch._foreach_add_(tab["steps"], 1.0)
        b1, b2 = g["betas"]
        with torch.cuda.device(params[0].device):
            rc = _lib.lib().gvl_clip_adam_step_f32(tab["descs"].data_ptr(), tab["n"], tab["cmap"].data_ptr(), tab["n_chunks"],
                                                   tab["partial"].data_ptr(), tab["scal"].data_ptr(), tab["corr"].data_ptr(),
                                                   self.max_norm, float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                                   float(g["weight_decay"]), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "clip_adam_step")
        bump_versions([r[0] for r in rows] + [r[1] for r in rows])          # parameters updated, gradients clipped in place
        self.last = tab["scal"]
        return True
