"""Hungarian matcher -- mirror of pdvc/matcher.py:53-150 (+ the 1-D box ops of misc/detr_utils/box_ops.py:8-47).

The cost matrix is built on the GPU with the reference's exact sequence of fp32 tensor ops (so that the matrix that
reaches the solver is the one the reference would produce on the same device), moved to the host in ONE copy, and
solved for all videos concurrently by the C++ solver of libgvl_msda.so (gvl_hungarian_batch_f32), which is
bit-identical to scipy.optimize.linear_sum_assignment (tests/test_abi.py).  Returns the reference's structure:
``(indices, rl_indices)``, each a list over videos of ``(int64 query ids, int64 gt ids)``; ``rl_indices`` is the
many-to-one assignment on the cost block tiled 4x (matcher.py:125-128).
"""
import ctypes

import numpy as np
import torch
from torch import nn

from . import _lib


def box_cl_to_xy(x):
    c, l = x.unbind(-1)
    return torch.stack([c - 0.5 * l, c + 0.5 * l], dim=-1)


def box_xy_to_cl(x):
    x0, x1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (x1 - x0)], dim=-1)


def box_iou(boxes1, boxes2):
    """1-D IoU and union, box_ops.py:19-27 (note the +1e-5 in the denominator)."""
    area1 = boxes1[:, 1] - boxes1[:, 0]
    area2 = boxes2[:, 1] - boxes2[:, 0]
    lt = torch.max(boxes1[:, None, 0], boxes2[:, 0])
    rb = torch.min(boxes1[:, None, 1], boxes2[:, 1])
    inter = (rb - lt).clamp(min=0)
    union = area1[:, None] + area2 - inter
    return inter / (union + 1e-5), union


def generalized_box_iou(boxes1, boxes2, check=True):
    """box_ops.py:30-47.  check=False skips the two degenerate-box asserts (each is a device->host sync); callers
    that skip them fold the same test into a flag that travels with their own device->host copy."""
    if check:
        assert (boxes1[:, 1:] >= boxes1[:, :1]).all()
        assert (boxes2[:, 1:] >= boxes2[:, :1]).all()
    iou, union = box_iou(boxes1, boxes2)
    lt = torch.min(boxes1[:, None, 0], boxes2[:, 0])
    rb = torch.max(boxes1[:, None, 1], boxes2[:, 1])
    area = (rb - lt).clamp(min=0)
    return iou - (area - union) / (area + 1e-5)


def hungarian_batch(C, sizes, m2o_rate=4, num_threads=0):
    """C: HOST float32 tensor (B, Q, sum(sizes)).  -> (indices, rl_indices) as lists of (rows, cols) int64 tensors."""
    C = C.contiguous()
    assert C.dtype == torch.float32 and C.device.type == "cpu"
    B, Q, G = C.shape
    sz = np.asarray(sizes, dtype=np.int32)
    n1 = [min(Q, int(n)) for n in sizes]
    n4 = [min(Q, int(n) * m2o_rate) for n in sizes]
    ir, ic = torch.empty(sum(n1), dtype=torch.int64), torch.empty(sum(n1), dtype=torch.int64)
    rr, rc = torch.empty(sum(n4), dtype=torch.int64), torch.empty(sum(n4), dtype=torch.int64)
    rcode = _lib.lib().gvl_hungarian_batch_f32(C.data_ptr(), B, Q, G, sz.ctypes.data_as(ctypes.c_void_p), m2o_rate,
                                               ir.data_ptr(), ic.data_ptr(), rr.data_ptr(), rc.data_ptr(),
                                               num_threads)
    if rcode != 0:
        raise ValueError("cost matrix is infeasible / contains invalid numeric entries")   # scipy's error
    indices = [(a, b) for a, b in zip(ir.split(n1), ic.split(n1))]
    rl = [(a, b) for a, b in zip(rr.split(n4), rc.split(n4))]
    return indices, rl


def lsap_launch(C, problems, max_rows, max_cols, rows, cols, status):
    """one solver launch over `problems` (a contiguous row block of a descriptor tensor) into existing result buffers"""
    with torch.cuda.device(C.device):
        rc = _lib.lib().gvl_lsap_batch_device_f32(C.data_ptr(), problems.data_ptr(), problems.shape[0], max_rows,
                                                  max_cols, rows.data_ptr(), cols.data_ptr(), status.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "lsap_batch_device")


def lsap_batch_device(C, problems, out_total, max_rows, max_cols, fill=None):
    """Solve assignment problems on the device (include/gvl_msda.h: gvl_lsap_batch_device_f32).
    C: float32 CUDA tensor (any shape; problems index its storage); problems: int64 CUDA tensor (n, 8) of
    {base, ld, Q, n, tile, out_off, 0, 0}.  -> (rows (out_total,), cols (out_total,), status (1,) int32)"""
    assert C.is_cuda and C.dtype == torch.float32 and C.is_contiguous()
    assert problems.is_cuda and problems.dtype == torch.int64 and problems.is_contiguous() and problems.shape[1] == 8
    if fill is None:
        rows = torch.empty(out_total, dtype=torch.int64, device=C.device)
        cols = torch.empty(out_total, dtype=torch.int64, device=C.device)
    else:                          # padded result slots: entries no problem writes keep the fill value
        rows = torch.full((2, out_total), fill, dtype=torch.int64, device=C.device)
        rows, cols = rows[0], rows[1]
    status = torch.zeros(1, dtype=torch.int32, device=C.device)
    lsap_launch(C, problems, max_rows, max_cols, rows, cols, status)
    return rows, cols, status


class MatchPlan:
    """Static (host-known) layout of one step's assignment problems: sizes of the GT sets are known on the host, so
    problem descriptors, output offsets and the per-entry bookkeeping (video of every match, row of its target in
    the concatenated targets) are built once per (layers, B, Q, sizes) and reused -- nothing here depends on data."""

    def __init__(self, nl, B, Q, sizes, device, m2o_rate):
        self.nl, self.B, self.Q, self.sizes, self.m2o = nl, B, Q, sizes, m2o_rate
        G = sum(sizes)
        coff = [0]
        for n in sizes[:-1]:
            coff.append(coff[-1] + n)
        self.n1 = [min(Q, n) for n in sizes]
        self.n4 = [min(Q, n * m2o_rate) for n in sizes]
        t1, t4 = sum(self.n1), sum(self.n4)
        self.t1, self.t4 = t1, t4
        desc = []
        for l in range(nl):
            o = l * t1
            for i, n in enumerate(sizes):
                desc.append([((l * B + i) * Q) * G + coff[i], G, Q, n, 1, o, 0, 0])
                o += self.n1[i]
        for l in range(nl):
            o = nl * t1 + l * t4
            for i, n in enumerate(sizes):
                desc.append([((l * B + i) * Q) * G + coff[i], G, Q, n, m2o_rate, o, 0, 0])
                o += self.n4[i]
        desc = [d for d in desc if d[3] > 0]                       # videos without GT have no problem
        self.n_one = sum(1 for d in desc if d[4] == 1)             # the one-to-one problems come first
        self.problems = torch.tensor(desc, dtype=torch.int64, device=device).reshape(-1, 8)
        self.out_total = nl * (t1 + t4)
        self.max_rows = max([1] + [min(Q, n * m2o_rate) for n in sizes])
        self.max_cols = max([1] + [max(Q, n * m2o_rate) for n in sizes])
        vid1 = [i for i, k in enumerate(self.n1) for _ in range(k)]
        self.vid_of_entry = torch.tensor(vid1, dtype=torch.int64, device=device)
        self.slot_of_entry = torch.tensor([k for n in self.n1 for k in range(n)], dtype=torch.int64, device=device)
        self.tgt_base = torch.tensor([coff[i] for i in vid1], dtype=torch.int64, device=device)
        self.cnt = torch.tensor(self.n1, dtype=torch.int64, device=device)
        starts = [0]
        for k in self.n1:
            starts.append(starts[-1] + k)
        self.ent_start = torch.tensor(starts, dtype=torch.int64, device=device)       # first pair of every video


class PaddedMatchPlan:
    """The same bookkeeping for ``gvl_amd.targets.PaddedTargets``: every video owns `slots` target slots and
    G1 = min(Q, slots) one-to-one / G4 = min(Q, 4 slots) many-to-one result slots, so nothing here depends on how
    many events a video really has -- the per-video counts are read on the device (column 3 of the problem
    descriptors is refreshed from ``targets.counts`` before every solve).  Result slots beyond a video's count hold
    -1.  This is what a layout-independent hipGraph capture of the step needs."""
    padded = True

    def __init__(self, nl, B, Q, slots, device, m2o_rate):
        self.nl, self.B, self.Q, self.slots, self.m2o = nl, B, Q, slots, m2o_rate
        self.G1, self.G4 = min(Q, slots), min(Q, slots * m2o_rate)
        self.t1, self.t4 = B * self.G1, B * self.G4
        desc = []
        for tile, per, first in ((1, self.G1, 0), (m2o_rate, self.G4, nl * self.t1)):
            for l in range(nl):
                for i in range(B):
                    desc.append([((l * B + i) * Q) * slots, slots, Q, 0, tile, first + (l * B + i) * per, 0, 0])
        self.problems = torch.tensor(desc, dtype=torch.int64, device=device).reshape(-1, 8)
        self.n_one = nl * B
        self.out_total = nl * (self.t1 + self.t4)
        self.max_rows, self.max_cols = min(Q, slots * m2o_rate), max(Q, slots * m2o_rate)
        ar = torch.arange(B, dtype=torch.int64, device=device)
        self.vid_of_entry = ar.repeat_interleave(self.G1)
        self.slot_of_entry = torch.arange(self.G1, dtype=torch.int64, device=device).repeat(B)
        self.tgt_base = self.vid_of_entry * slots
        self.ent_start = torch.arange(B + 1, dtype=torch.int64, device=device) * self.G1

    def refresh(self, counts):
        """write n = counts[video] into every problem descriptor (one small device copy; capturable)"""
        self.problems.view(2 * self.nl, self.B, 8)[:, :, 3] = counts
        self.pair_count = counts.clamp(max=self.Q)                      # matches per video, one-to-one
        self.valid = self.slot_of_entry < self.pair_count[self.vid_of_entry]


class ManyToOne:
    """The 4x-tiled many-to-one assignment (matcher.py:125-128), solved ON DEMAND.  The reference computes it in every
    matcher call and returns it as the second element of `indices`, but nothing on its path reads it (criterion.py:52,108
    unpack and drop it; pdvc.py:419,524,617,647 take `[0]`), and its problems are 4x taller -- ~90 % of the solver's time.
    Here the problems (same cost tensor, same descriptors) are solved the first time someone asks --
    ``LayerMatch[1]`` / ``.rl_q`` / ``.rl_t`` -- with the same kernel, into the same result buffers, bit-identical."""

    def __init__(self, C, plan, rows, cols, status):
        self.C, self.plan, self.rows, self.cols, self.status = C, plan, rows, cols, status
        self.done = False

    @torch.no_grad()
    def ensure(self):
        if not self.done:
            p_ = self.plan
            if p_.problems.shape[0] > p_.n_one:
                lsap_launch(self.C, p_.problems[p_.n_one:], p_.max_rows, p_.max_cols, self.rows, self.cols, self.status)
            self.done = True


class LayerMatch:
    """The matching of one decoder layer, resident on the device.  ``q`` / ``t``: matched query id and (video-local)
    target id of every match, videos concatenated; ``t_global`` the target's row in the concatenated targets.
    Indexing ([0] -> indices, [1] -> rl_indices) materialises the reference's host structure (one copy)."""

    def __init__(self, plan, layer, rows, cols, status, ok, targets=None, many=None):
        self.plan, self.layer, self.targets = plan, layer, targets      # targets: PaddedTargets of a padded plan
        a = layer * plan.t1
        self.q, self.t = rows[a:a + plan.t1], cols[a:a + plan.t1]
        b = plan.nl * plan.t1 + layer * plan.t4
        self._rl = (rows[b:b + plan.t4], cols[b:b + plan.t4])
        self.many = many                                                # ManyToOne shared by the layers of one step
        self.status, self.ok = status, ok
        self.rows_all, self.cols_all = rows, cols       # all layers, for consumers that process the layers together
        self._host = None

    @property
    def rl_q(self):
        if self.many is not None:
            self.many.ensure()
        return self._rl[0]

    @property
    def rl_t(self):
        if self.many is not None:
            self.many.ensure()
        return self._rl[1]

    def invalidate(self):
        """the device buffers behind this match were rewritten (a hipGraph replay): drop everything derived from them"""
        self._host = None
        if self.many is not None:
            self.many.done = False

    @property
    def t_global(self):
        return self.t + self.plan.tgt_base

    def check(self):
        """raises what the reference raises (scipy ValueError / box_ops assert); costs one tiny device->host read"""
        if int(self.status) != 0:
            raise ValueError("cost matrix is infeasible / contains invalid numeric entries")
        assert bool(self.ok.all()), "degenerate boxes (x1 < x0) in the matcher"

    def _split(self, q, t_, per, tile):
        """flat result slots -> the reference's per-video (query ids, target ids) lists"""
        p_ = self.plan
        q, t_ = q.cpu(), t_.cpu()
        if getattr(p_, "padded", False):     # padded result slots; the counts of the loaded batch are known on the host
            ns = [min(p_.Q, n * tile) for n in self.targets.host_counts]
            return [(q.view(p_.B, per)[i, :n], t_.view(p_.B, per)[i, :n]) for i, n in enumerate(ns)]
        sizes = p_.n1 if tile == 1 else p_.n4
        return [(a, b) for a, b in zip(q.split(sizes), t_.split(sizes))]

    def host(self, kind=None):
        """the reference's host structure (matcher.py:124-131): kind 0 -> indices, 1 -> rl_indices (solved on demand),
        None -> both"""
        if self._host is None:
            self.check()
            self._host = [None, None]
        p_ = self.plan
        if kind in (0, None) and self._host[0] is None:
            self._host[0] = self._split(self.q, self.t, getattr(p_, "G1", 0), 1)
        if kind in (1, None) and self._host[1] is None:
            self._host[1] = self._split(self.rl_q, self.rl_t, getattr(p_, "G4", 0), p_.m2o)
        return self._host

    def __getitem__(self, k):
        return self.host(k)[k]

    def __iter__(self):
        return iter(self.host())

    def __len__(self):
        return 2


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1, cost_alpha=0.25,
                 cost_gamma=2, cost_cl=0, opt=None):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        self.cost_alpha, self.cost_gamma, self.cost_cl = cost_alpha, cost_gamma, cost_cl
        self.opt = opt
        self._plans = {}

    @torch.no_grad()
    def cost_matrix(self, outputs, targets, tgt_cat=None, with_flag=False):
        """matcher.py:74-105 -> C (B, Q, sum nGT) on the device of the predictions.  with_flag=True returns
        (C, ok) where ok is the device-side result of box_ops.py:39-40's degenerate-box asserts."""
        bs, num_queries = outputs["pred_logits"].shape[:2]
        out_prob = outputs["pred_logits"].flatten(0, 1).sigmoid()
        out_bbox = outputs["pred_boxes"].flatten(0, 1)
        if tgt_cat is None:
            tgt_cat = (torch.cat([v["labels"] for v in targets]), torch.cat([v["boxes"] for v in targets]))
        tgt_ids, tgt_bbox = tgt_cat
        alpha, gamma = self.cost_alpha, self.cost_gamma
        neg = (1 - alpha) * (out_prob ** gamma) * (-(1 - out_prob + 1e-8).log())
        pos = alpha * ((1 - out_prob) ** gamma) * (-(out_prob + 1e-8).log())
        cost_class = pos[:, tgt_ids] - neg[:, tgt_ids]
        # every reference config sets set_cost_bbox = 0; x + 0 * finite == x bit for bit, so the (4800 x nGT) L1 distance
        # matrix is only built when it can influence the cost (matcher.py:85 computes it unconditionally)
        cost_bbox = torch.cdist(out_bbox, tgt_bbox, p=1) if self.cost_bbox != 0 else 0.0
        xy1, xy2 = box_cl_to_xy(out_bbox), box_cl_to_xy(tgt_bbox)
        cost_giou = -generalized_box_iou(xy1, xy2, check=not with_flag)
        cl = outputs.get('cl_match_mats', 0)
        cost_cl = -1.0 * cl[:, :tgt_bbox.shape[0]] if isinstance(cl, torch.Tensor) else -1 * cl
        C = self.cost_bbox * cost_bbox + self.cost_class * cost_class + self.cost_giou * cost_giou \
            + self.cost_cl * cost_cl
        if self.opt is not None and getattr(self.opt, "set_cost_caption", 0) > 0 and 'cap_cost_mat' in outputs:
            C = C + self.opt.set_cost_caption * outputs['cap_cost_mat']
        C = C.view(bs, num_queries, -1)
        if with_flag:
            ok = (xy1[:, 1:] >= xy1[:, :1]).all() & (xy2[:, 1:] >= xy2[:, :1]).all()
            return C, ok
        return C

    # ---- on-device matching (no device->host copy; capturable in a hipGraph) ----------------------------------
    def _plan(self, nl, B, Q, sizes, device, m2o_rate=4):
        key = (nl, B, Q, tuple(sizes), str(device), m2o_rate)
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) >= 64:          # one plan per batch LAYOUT: bounded (a real loop rarely repeats a layout)
                self._plans.pop(next(iter(self._plans)))
            plan = MatchPlan(nl, B, Q, list(sizes), device, m2o_rate)
            self._plans[key] = plan
        return plan

    @torch.no_grad()
    def match_layers_device(self, outputs_list, targets):
        """All decoder layers x videos x {one-to-one, 4x-tiled} assignment problems of a step in ONE kernel launch on
        the device (gvl_lsap_batch_device_f32; bit-identical to scipy).  Returns one LayerMatch per layer whose
        index tensors stay on the device; LayerMatch[0] / [1] materialise the reference's (indices, rl_indices)."""
        sizes = [len(v["boxes"]) for v in targets]
        tgt_cat = (torch.cat([v["labels"] for v in targets]), torch.cat([v["boxes"] for v in targets]))
        C, ok = self.cost_matrices(outputs_list, targets, tgt_cat)    # (nl, B, Q, G)
        nl, B, Q, G = C.shape
        plan = self._plan(nl, B, Q, sizes, C.device)
        if plan.max_rows > self.LSAP_DEVICE_MAX_ROWS or plan.max_cols > self.LSAP_DEVICE_MAX_COLS:
            return None                                   # beyond the on-chip solver: caller takes the host path
        rows, cols, status = lsap_batch_device(C, plan.problems[:plan.n_one], plan.out_total, plan.max_rows,
                                               plan.max_cols)
        many = ManyToOne(C, plan, rows, cols, status)               # the 4x-tiled problems: solved when first read
        return [LayerMatch(plan, l, rows, cols, status, ok, many=many) for l in range(nl)]

    LSAP_DEVICE_MAX_ROWS, LSAP_DEVICE_MAX_COLS = 256, 1024        # on-chip limits of k_lsap (gvl_lsap_dev.hip)

    def padded_eligible(self, outputs_list, Q, slots, m2o_rate=4):
        """domain of the layout-independent path: fused cost kernel (class / box / GIoU terms only, fp32, on the GPU)
        and assignment problems that fit the on-device solver"""
        first = outputs_list[0]
        extra = any(isinstance(o.get('cl_match_mats', 0), torch.Tensor) and self.cost_cl != 0 for o in outputs_list) \
            or (self.opt is not None and getattr(self.opt, "set_cost_caption", 0) > 0
                and any('cap_cost_mat' in o for o in outputs_list))
        return (not extra and first["pred_logits"].is_cuda
                and min(Q, slots * m2o_rate) <= self.LSAP_DEVICE_MAX_ROWS
                and max(Q, slots * m2o_rate) <= self.LSAP_DEVICE_MAX_COLS)

    @torch.no_grad()
    def match_layers_padded(self, outputs_list, pt, m2o_rate=4):
        """match_layers_device on ``PaddedTargets``: cost blocks (nl, B, Q, slots) from gvl_match_cost_padded_f32, one
        solver launch over 2 * nl * B fixed problem slots whose sizes come from ``pt.counts`` on the device.  No
        tensor shape, launch parameter or host value depends on the number of events per video."""
        logits = torch.stack([o["pred_logits"] for o in outputs_list]).float().contiguous()
        boxes = torch.stack([o["pred_boxes"] for o in outputs_list]).float().contiguous()
        nl, B, Q, NC = logits.shape
        key = ("padded", nl, B, Q, pt.slots, str(logits.device), m2o_rate)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = PaddedMatchPlan(nl, B, Q, pt.slots, logits.device, m2o_rate)
        plan.refresh(pt.counts)
        C = torch.empty((nl, B, Q, pt.slots), dtype=torch.float32, device=logits.device)
        ok = torch.ones(1, dtype=torch.int32, device=logits.device)
        with torch.cuda.device(logits.device):
            rc = _lib.lib().gvl_match_cost_padded_f32(
                logits.data_ptr(), boxes.data_ptr(), pt.labels.data_ptr(), pt.boxes.data_ptr(), pt.counts.data_ptr(),
                nl, B, Q, NC, pt.slots, float(self.cost_class), float(self.cost_bbox), float(self.cost_giou),
                float(self.cost_alpha), float(self.cost_gamma), C.data_ptr(), ok.data_ptr(),
                torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "match_cost_padded")
        rows, cols, status = lsap_batch_device(C, plan.problems[:plan.n_one], plan.out_total, plan.max_rows,
                                               plan.max_cols, fill=-1)
        many = ManyToOne(C, plan, rows, cols, status)               # the 4x-tiled problems: solved when first read
        return [LayerMatch(plan, l, rows, cols, status, ok, targets=pt, many=many) for l in range(nl)]

    @torch.no_grad()
    def cost_matrices(self, outputs_list, targets, tgt_cat, fused=None):
        """cost tensors of all decoder layers (nl, B, Q, G) + the degenerate-box flag.  One HIP launch
        (gvl_match_cost_f32) when only the class / box / GIoU terms contribute; the PyTorch op sequence of
        ``cost_matrix`` per layer otherwise (contrastive / caption cost terms)."""
        first = outputs_list[0]
        fused = getattr(self, "fused_cost", True) if fused is None else fused
        extra = any(isinstance(o.get('cl_match_mats', 0), torch.Tensor) and self.cost_cl != 0 for o in outputs_list) \
            or (self.opt is not None and getattr(self.opt, "set_cost_caption", 0) > 0
                and any('cap_cost_mat' in o for o in outputs_list))
        if fused and not extra and first["pred_logits"].is_cuda and first["pred_logits"].dtype == torch.float32:
            logits = torch.stack([o["pred_logits"] for o in outputs_list]).contiguous()
            boxes = torch.stack([o["pred_boxes"] for o in outputs_list]).contiguous()
            nl, B, Q, NC = logits.shape
            labels, tboxes = tgt_cat[0].contiguous(), tgt_cat[1].float().contiguous()
            G = labels.shape[0]
            C = torch.empty((nl, B, Q, G), dtype=torch.float32, device=logits.device)
            ok = torch.ones(1, dtype=torch.int32, device=logits.device)
            with torch.cuda.device(logits.device):
                rc = _lib.lib().gvl_match_cost_f32(
                    logits.data_ptr(), boxes.data_ptr(), labels.data_ptr(), tboxes.data_ptr(), nl, B, Q, NC, G,
                    float(self.cost_class), float(self.cost_bbox), float(self.cost_giou), float(self.cost_alpha),
                    float(self.cost_gamma), C.data_ptr(), ok.data_ptr(), torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "match_cost")
            return C, ok
        Cs, oks = zip(*[self.cost_matrix(o, targets, tgt_cat, with_flag=True) for o in outputs_list])
        return torch.stack(Cs).float().contiguous(), torch.stack(oks).all()

    @torch.no_grad()
    def match_layers(self, outputs_list, targets):
        """Host variant: all cost matrices cross PCIe in ONE copy (with the degenerate-box flag) and are solved by the
        C++ solver.  Returns [(indices, rl_indices)] per layer, identical to calling forward() per layer."""
        sizes = [len(v["boxes"]) for v in targets]
        tgt_cat = (torch.cat([v["labels"] for v in targets]), torch.cat([v["boxes"] for v in targets]))
        Cs, oks = zip(*[self.cost_matrix(o, targets, tgt_cat, with_flag=True) for o in outputs_list])
        nl = len(Cs)
        B, Q, G = Cs[0].shape
        packed = torch.cat([torch.stack(Cs).float().reshape(-1), torch.stack(oks).float()]).cpu()
        assert bool(packed[-nl:].all()), "degenerate boxes (x1 < x0) in the matcher"          # box_ops.py:39-40
        C = packed[:-nl].view(nl * B, Q, G)
        return [hungarian_batch(C[l * B:(l + 1) * B], sizes, m2o_rate=4) for l in range(nl)]

    @torch.no_grad()
    def forward(self, outputs, targets, verbose=False, return_C=False):
        C = self.cost_matrix(outputs, targets).float().cpu()          # the single device -> host copy (matcher.py:120)
        sizes = [len(v["boxes"]) for v in targets]
        indices, rl_indices = hungarian_batch(C, sizes, m2o_rate=4)
        if return_C:
            return indices, rl_indices, [c[i] for i, c in enumerate(C.split(sizes, -1))]
        return indices, rl_indices


def build_matcher(args):
    return HungarianMatcher(cost_class=args.set_cost_class, cost_bbox=args.set_cost_bbox,
                            cost_giou=args.set_cost_giou, cost_alpha=args.cost_alpha, cost_gamma=args.cost_gamma,
                            cost_cl=vars(args).get('set_cost_cl', 0.), opt=args)
