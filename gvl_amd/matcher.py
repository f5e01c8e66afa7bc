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


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1, cost_alpha=0.25,
                 cost_gamma=2, cost_cl=0, opt=None):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        self.cost_alpha, self.cost_gamma, self.cost_cl = cost_alpha, cost_gamma, cost_cl
        self.opt = opt

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
        cost_bbox = torch.cdist(out_bbox, tgt_bbox, p=1)
        xy1, xy2 = box_cl_to_xy(out_bbox), box_cl_to_xy(tgt_bbox)
        cost_giou = -generalized_box_iou(xy1, xy2, check=not with_flag)
        cl = outputs.get('cl_match_mats', 0)
        cost_cl = -1.0 * cl[:, :cost_bbox.shape[1]] if isinstance(cl, torch.Tensor) else -1 * cl
        C = self.cost_bbox * cost_bbox + self.cost_class * cost_class + self.cost_giou * cost_giou \
            + self.cost_cl * cost_cl
        if self.opt is not None and getattr(self.opt, "set_cost_caption", 0) > 0 and 'cap_cost_mat' in outputs:
            C = C + self.opt.set_cost_caption * outputs['cap_cost_mat']
        C = C.view(bs, num_queries, -1)
        if with_flag:
            ok = (xy1[:, 1:] >= xy1[:, :1]).all() & (xy2[:, 1:] >= xy2[:, :1]).all()
            return C, ok
        return C

    @torch.no_grad()
    def match_layers(self, outputs_list, targets):
        """The matcher for several decoder layers at once (criterion.py:173 + :192 call it once per layer, each with
        its own device->host copy): all cost matrices are computed on the device, cross PCIe in ONE copy together
        with the degenerate-box flag, and all layers x videos are solved in one batch of host threads.
        Returns [(indices, rl_indices)] per layer, identical to calling forward() per layer."""
        sizes = [len(v["boxes"]) for v in targets]
        tgt_cat = (torch.cat([v["labels"] for v in targets]), torch.cat([v["boxes"] for v in targets]))
        Cs, oks = zip(*[self.cost_matrix(o, targets, tgt_cat, with_flag=True) for o in outputs_list])
        nl = len(Cs)
        B, Q, G = Cs[0].shape
        packed = torch.cat([torch.stack(Cs).float().reshape(-1), torch.stack(oks).float()]).cpu()
        assert bool(packed[-nl:].all()), "degenerate boxes (x1 < x0) in the matcher"          # box_ops.py:39-40
        C = packed[:-nl].view(nl * B, Q, G)
        # layer l, video i owns the same column block as video i: present it as nl*B "videos" over nl copies of the
        # column blocks by solving per layer (the C ABI takes one (B,Q,G) tensor per call; threads cover B)
        res = []
        for l in range(nl):
            res.append(hungarian_batch(C[l * B:(l + 1) * B], sizes, m2o_rate=4))
        return res

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
