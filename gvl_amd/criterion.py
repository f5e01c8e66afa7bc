"""Set criterion of the train step -- mirror of pdvc/criterion.py:16-257 (labels / boxes / cardinality losses).

Host-side PyTorch on the GPU tensors (elementwise work on <= B*Q*few values, kept out of
hand-kernel scope).  The only collective on the reference's hot path lives here: ``all_reduce(num_boxes)``
(criterion.py:178-180); it runs over RCCL when a process group exists (data-parallel training, gvl_amd.parallel).
The contrastive criterion (frozen RoBERTa text branch) is outside the accelerated path and is not built.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from . import _lib
from .matcher import box_cl_to_xy, box_iou

# Empirical distribution of the number of events per video used to re-weight the counter loss.
# DATA taken verbatim from the reference (pdvc/criterion.py:39-45); it is a dataset statistic, not code.
COUNTER_CLASS_RATE = [
    0.00000000e+00, 0.00000000e+00, 1.93425917e-01, 4.12129084e-01, 1.88929963e-01, 7.81296833e-02, 5.09541413e-02,
    3.12718553e-02, 1.84833650e-02, 8.39244680e-03, 6.59406534e-03, 4.49595364e-03, 2.19802178e-03, 1.79838146e-03,
    5.99460486e-04, 4.99550405e-04, 4.99550405e-04, 1.99820162e-04, 2.99730243e-04, 3.99640324e-04, 2.99730243e-04,
    0.00000000e+00, 1.99820162e-04, 0.00000000e+00, 0.00000000e+00, 0.00000000e+00, 9.99100809e-05, 9.99100809e-05]


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def sigmoid_focal_loss(inputs, targets, num_boxes, alpha: float = 0.25, gamma: float = 2):
    """criterion.py:232-257"""
    prob = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum() / num_boxes


def cross_entropy_with_gaussian_mask(inputs, targets, opt, weight):
    """criterion.py:209-229: BCE on the event-count one-hot, off-target bins damped by a gaussian around the target."""
    n, width = targets.shape
    idx = torch.arange(width, device=inputs.device).float()
    mask_dict = torch.exp(-(idx[:, None] - idx[None, :]) ** 2 / (2 * 2 ** 2))
    mask = mask_dict[targets.max(dim=1)[1]]
    loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none", weight=1 - weight)
    if opt.lloss_gau_mask:
        coef = targets + ((1 - mask) ** opt.lloss_beta) * (1 - targets)
    else:
        coef = targets + (1 - targets)
    return (loss * coef).mean(1).mean()


LOSS_KEYS = ('loss_ce', 'loss_counter', 'loss_bbox', 'loss_giou', 'loss_self_iou', 'cardinality_error')


class SetCriterionFunction(torch.autograd.Function):
    """labels / boxes / cardinality losses of ALL decoder layers as one autograd node on two HIP launches
    (include/gvl_msda.h: gvl_set_criterion_forward_f32 / _backward_f32) -> losses (n_layers, 6) in LOSS_KEYS order.
    The PyTorch formulation below (SetCriterion.loss_*) is ~190 launch-bound kernels per layer and direction."""

    @staticmethod
    def _call(fn, ctx_t, cfg, padded, *outs):
        logits, counts, boxes, mq, mt, vid, tbase, ent_start, labels, tboxes, gt_counts, ccr = ctx_t[:12]
        nl, B, Q, NC = logits.shape
        pair_count = ctx_t[12].data_ptr() if padded else None
        nb_dev = ctx_t[-1].data_ptr() if len(ctx_t) > (13 if padded else 12) else None
        with torch.cuda.device(logits.device):
            rc = fn(logits.data_ptr(), counts.data_ptr(), boxes.data_ptr(), mq.data_ptr(), mt.data_ptr(),
                    vid.data_ptr(), tbase.data_ptr(), ent_start.data_ptr(), labels.data_ptr(), tboxes.data_ptr(),
                    gt_counts.data_ptr(), ccr.data_ptr(), nl, B, Q, NC, counts.shape[-1], mq.shape[-1],
                    labels.shape[0], *cfg, pair_count, nb_dev, *[o.data_ptr() for o in outs],
                    torch.cuda.current_stream().cuda_stream)
        return rc

    @staticmethod
    def forward(ctx, logits, counts, boxes, mq, mt, vid, tbase, ent_start, labels, tboxes, gt_counts, ccr, num_boxes,
                alpha, gamma, beta, gau_mask, pair_count=None, num_boxes_dev=None):
        """pair_count (B): the padded, layout-independent form (gvl_amd.targets.PaddedTargets) -- match slots beyond a
        video's count are empty.  num_boxes_dev (1): the normaliser of criterion.py:178-181 read from device memory
        (padded form; or a captured list-form step whose cross-rank mean changes from batch to batch)."""
        tensors = (logits.contiguous(), counts.contiguous(), boxes.contiguous(), mq.contiguous(), mt.contiguous(), vid,
                   tbase, ent_start, labels.contiguous(), tboxes.float().contiguous(), gt_counts, ccr)
        padded = pair_count is not None
        if padded:
            tensors = tensors + (pair_count.contiguous(),)
        if num_boxes_dev is not None:
            tensors = tensors + (num_boxes_dev,)
        cfg = (1.0 if num_boxes_dev is not None else float(num_boxes), float(alpha), float(gamma), float(beta),
               int(bool(gau_mask)))
        losses = torch.empty((logits.shape[0], len(LOSS_KEYS)), dtype=torch.float32, device=logits.device)
        _lib.check(SetCriterionFunction._call(_lib.lib().gvl_set_criterion_forward_f32, tensors, cfg, padded, losses),
                   "set_criterion_forward")
        ctx.save_for_backward(*tensors)
        ctx.cfg, ctx.padded = cfg, padded
        return losses

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_losses):
        tensors = ctx.saved_tensors
        g_logits, g_counts, g_boxes = (torch.empty_like(t_) for t_ in tensors[:3])
        _lib.check(SetCriterionFunction._call(_lib.lib().gvl_set_criterion_backward_f32, tensors, ctx.cfg, ctx.padded,
                                              grad_losses.contiguous(), g_logits, g_counts, g_boxes),
                   "set_criterion_backward")
        return (g_logits, g_counts, g_boxes) + (None,) * 16


def unbind_tagged(vec):
    """vec.unbind(0) whose scalars remember where they came from (`_gvl_vec` = (vec, index)): a weighted sum of loss terms can then
    be ONE dot product with the vector itself (gvl_amd.parallel.TrainStep._forward_loss) -- autograd sees no unbind, whose backward
    is a zeros() per unused entry and a stack"""
    parts = vec.unbind(0)
    for i, t_ in enumerate(parts):
        t_._gvl_vec = (vec, i)
    return parts


def weighted_loss_sum(loss, weight_dict, cache=None):
    """train.py:403 -- `sum(loss[k] * weight_dict[k] for k in loss.keys() if k in weight_dict)` -- for scalar tensor terms as ONE dot
    product: 2-3 launches forward and one backward instead of a mul + add per term and their backward (30 launches for the 10
    terms of cfg A).  Terms that are entries of a loss VECTOR (unbind_tagged: the criterion's table, the captioner's per-layer
    losses) are taken from the vector itself -- dot(cat(vectors)[weighted entries], weights) --, so autograd sees no unbind; the
    rest are stacked.  The unweighted entries are SELECTED away, never multiplied by 0: loss_self_iou is 0/0 for a single match,
    as the reference's.  cache: a dict the caller keeps (weights / indices per term layout AND weight values)."""
    wd = weight_dict
    keys = tuple(k for k in loss.keys() if k in wd)
    if not (len(keys) > 2 and all(isinstance(loss[k], torch.Tensor) and loss[k].dim() == 0 for k in keys)
            and len({loss[k].device for k in keys}) == 1):
        return sum(loss[k].float() * wd[k] for k in keys)
    cache = {} if cache is None else cache
    dev = loss[keys[0]].device
    vecs, offs, where, loose, total = [], {}, [], [], 0
    for k in keys:
        tag = getattr(loss[k], "_gvl_vec", None)
        if tag is None or tag[0].dim() != 1 or tag[0].dtype != torch.float32 or tag[0].device != dev:
            loose.append(k)
            continue
        if id(tag[0]) not in offs:
            offs[id(tag[0])] = total
            vecs.append(tag[0])
            total += tag[0].numel()
        where.append((k, offs[id(tag[0])] + tag[1]))
    if loose:
        vecs.append(torch.stack([loss[k].float() for k in loose]))
        where += [(k, total + i) for i, k in enumerate(loose)]
        total += len(loose)
    weights = tuple(float(wd[k]) for k, _ in where)
    sig = (keys, tuple(v.numel() for v in vecs), tuple(i for _, i in where), weights, str(dev))
    hit = cache.get(sig)
    if hit is None:
        hit = cache[sig] = (torch.tensor(weights, dtype=torch.float32, device=dev),
                            torch.tensor([i for _, i in where], dtype=torch.int64, device=dev))
    full = vecs[0] if len(vecs) == 1 else torch.cat(vecs)
    if [i for _, i in where] != list(range(total)):
        full = full.index_select(0, hit[1])
    return torch.dot(full, hit[0])


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, losses, focal_alpha=0.25, focal_gamma=2, opt={}):
        super().__init__()
        self.num_classes = num_classes
        self.matcher = matcher
        self.weight_dict = weight_dict
        self.losses = losses
        self.focal_alpha = focal_alpha
        self.focal_gamma = focal_gamma
        self.opt = opt
        self.counter_class_rate = torch.tensor(COUNTER_CLASS_RATE)
        self.device_matching = True      # solve the Hungarian problems on the GPU (bit-identical to scipy)
        self.fused = True                # all layers' losses in one HIP launch per direction (SetCriterionFunction)
        # set by a caller that has already taken the cross-rank mean (graph capture): a float, or a (1,) fp32 DEVICE
        # tensor the captured kernels read at replay time (so the value is not baked into the graph)
        self.num_boxes_override = None
        self._const_cache = {}

    @staticmethod
    def _pack(indices, targets, device):
        """All the index bookkeeping of one layer's matching, moved to the device in ONE copy:
        batch index and query index of every matched prediction (criterion.py:139-143), the row of its target in
        the concatenated target tensors, and the per-video match counts (for the self-IoU normaliser)."""
        sizes_t = [len(t_["labels"]) for t_ in targets]
        offs = [0]
        for n in sizes_t[:-1]:
            offs.append(offs[-1] + n)
        b = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        q = torch.cat([src for (src, _) in indices])
        tg = torch.cat([J + offs[i] for i, (_, J) in enumerate(indices)])
        cnt = torch.tensor([len(src) for (src, _) in indices], dtype=torch.int64)
        packed = torch.cat([b, q, tg, cnt]).to(device, non_blocking=True)
        n = b.numel()
        return packed[:n], packed[n:2 * n], packed[2 * n:3 * n], packed[3 * n:]

    def _layer_state(self, outputs, targets, indices):
        """device-side index tensors + concatenated targets, computed once per layer and shared by the losses"""
        key = id(indices)
        st = self._state.get(key)
        if st is None:
            dev = outputs['pred_logits'].device
            if hasattr(indices, "plan"):                               # LayerMatch: already on the device
                b, q, tg, cnt = indices.plan.vid_of_entry, indices.q, indices.t_global, indices.plan.cnt
            else:
                one2one, _ = indices
                b, q, tg, cnt = self._pack(one2one, targets, dev)
            st = {"b": b, "q": q, "tg": tg, "cnt": cnt, "labels": self._tgt_cat[0], "boxes": self._tgt_cat[1]}
            self._state[key] = st
        return st

    def loss_labels(self, outputs, targets, indices, num_boxes, log=True):
        st = self._layer_state(outputs, targets, indices)
        logits = outputs['pred_logits']
        dev = logits.device
        classes = torch.full(logits.shape[:2], self.num_classes, dtype=torch.int64, device=dev)
        classes[st["b"], st["q"]] = st["labels"][st["tg"]]
        onehot = torch.zeros([logits.shape[0], logits.shape[1], logits.shape[2] + 1], dtype=logits.dtype, device=dev)
        onehot.scatter_(2, classes.unsqueeze(-1), 1)
        onehot = onehot[:, :, :-1]
        losses = {'loss_ce': sigmoid_focal_loss(logits, onehot, num_boxes, alpha=self.focal_alpha,
                                                gamma=self.focal_gamma) * logits.shape[1]}
        pred_count = outputs['pred_count']
        max_length = pred_count.shape[1] - 1
        cnt_onehot = torch.zeros_like(pred_count)
        cnt_onehot.scatter_(1, self._gt_counts.clamp(max=max_length).unsqueeze(-1), 1)
        wkey = ("ccr", max_length, str(dev))
        weight = self._const_cache.get(wkey)
        if weight is None:
            weight = self._const_cache[wkey] = self.counter_class_rate[:max_length + 1].to(dev)
        losses['loss_counter'] = cross_entropy_with_gaussian_mask(pred_count, cnt_onehot, self.opt, weight)
        return losses

    @torch.no_grad()
    def loss_cardinality(self, outputs, targets, indices, num_boxes):
        logits = outputs['pred_logits']
        card_pred = (logits.argmax(-1) != logits.shape[-1] - 1).sum(1)
        return {'cardinality_error': F.l1_loss(card_pred.float(), self._gt_counts.float())}

    def loss_boxes(self, outputs, targets, indices, num_boxes):
        st = self._layer_state(outputs, targets, indices)
        src = outputs['pred_boxes'][st["b"], st["q"]]
        tgt = st["boxes"][st["tg"]]
        losses = {'loss_bbox': F.l1_loss(src, tgt, reduction='none').sum() / num_boxes}
        sxy, txy = box_cl_to_xy(src), box_cl_to_xy(tgt)
        # diagonal of the pairwise GIoU (criterion.py:117-119) computed pair by pair instead of n x n then diag
        inter = (torch.min(sxy[:, 1], txy[:, 1]) - torch.max(sxy[:, 0], txy[:, 0])).clamp(min=0)
        union = (sxy[:, 1] - sxy[:, 0]) + (txy[:, 1] - txy[:, 0]) - inter
        area = (torch.max(sxy[:, 1], txy[:, 1]) - torch.min(sxy[:, 0], txy[:, 0])).clamp(min=0)
        giou = inter / (union + 1e-5) - (area - union) / (area + 1e-5)
        losses['loss_giou'] = (1 - giou).sum() / num_boxes
        # self-IoU between the matched predictions of the same video (criterion.py:123-130): upper triangle of the
        # per-video block, normalised by the number of pairs s(s-1)/2 (0/0 = nan for s = 1, as in the reference)
        iou = box_iou(sxy, sxy)[0]
        same = st["b"][:, None] == st["b"][None, :]
        upper = torch.ones_like(iou, dtype=torch.bool).triu(diagonal=1)
        per_pred = (iou * (same & upper)).sum(1)
        per_video = torch.zeros(st["cnt"].numel(), dtype=iou.dtype, device=iou.device).index_add_(0, st["b"], per_pred)
        cntf = st["cnt"].to(iou.dtype)
        losses['loss_self_iou'] = (per_video / (0.5 * cntf * (cntf - 1))).sum()
        return losses

    def _fused_losses(self, layers, matches, num_boxes):
        """every loss term of every decoder layer from ONE autograd node (SetCriterionFunction)"""
        plan = matches[0].plan
        nl = len(layers)
        # (.float(): a no-op in fp32; under autocast the heads emit bf16 and the criterion kernels are an fp32 island)
        logits = torch.stack([o['pred_logits'] for o in layers]).float()
        counts = torch.stack([o['pred_count'] for o in layers]).float()
        boxes = torch.stack([o['pred_boxes'] for o in layers]).float()
        rows, cols = matches[0].rows_all, matches[0].cols_all           # (nl * t1 + ...) laid out layer-major
        mq, mt = rows[:nl * plan.t1].view(nl, plan.t1), cols[:nl * plan.t1].view(nl, plan.t1)
        dev = logits.device
        max_length = counts.shape[-1] - 1
        wkey = ("ccr", max_length, str(dev))
        ccr = self._const_cache.get(wkey)
        if ccr is None:
            ccr = self._const_cache[wkey] = self.counter_class_rate[:max_length + 1].to(dev)
        padded = getattr(plan, "padded", False)
        table = SetCriterionFunction.apply(
            logits, counts, boxes, mq, mt, plan.vid_of_entry, plan.tgt_base, plan.ent_start, self._tgt_cat[0],
            self._tgt_cat[1], self._gt_counts, ccr, num_boxes, self.focal_alpha, self.focal_gamma,
            getattr(self.opt, "lloss_beta", 1), getattr(self.opt, "lloss_gau_mask", 1),
            plan.pair_count if padded else None, num_boxes if isinstance(num_boxes, torch.Tensor) else None)
        flat = unbind_tagged(table.flatten())
        losses = {}
        for l in range(nl):
            suffix = "" if l == 0 else f"_{l - 1}"
            for k, name in enumerate(LOSS_KEYS):
                losses[name + suffix] = flat[l * len(LOSS_KEYS) + k]
        return losses

    def get_loss(self, loss, outputs, targets, indices, num_boxes, **kwargs):
        table = {'labels': self.loss_labels, 'cardinality': self.loss_cardinality, 'boxes': self.loss_boxes}
        assert loss in table, f'do you really want to compute {loss} loss?'
        return table[loss](outputs, targets, indices, num_boxes, **kwargs)

    def padded_static_ok(self, batch, queries, slots, m2o_rate=4):
        """the conditions of the layout-independent form that are known BEFORE a forward (what a captured step checks
        when it decides between one padded graph and one graph per batch layout -- ADVICE r2: the decision and
        `padded_eligible` below must agree, or a graph bakes one batch's event layout in)"""
        m = self.matcher
        return (self.fused and self.device_matching and set(self.losses) == {'labels', 'boxes', 'cardinality'}
                and batch * queries <= 24576 and hasattr(m, "match_layers_padded")
                and not (m.opt is not None and getattr(m.opt, "set_cost_caption", 0) > 0)
                and min(queries, slots * m2o_rate) <= m.LSAP_DEVICE_MAX_ROWS
                and max(queries, slots * m2o_rate) <= m.LSAP_DEVICE_MAX_COLS)

    def padded_eligible(self, outputs, pt):
        """can this step run in the layout-independent form (PaddedTargets in, everything on the device)?"""
        lg = outputs['pred_logits']
        layers = [outputs] + list(outputs.get('aux_outputs', []))
        return (lg.is_cuda and self.padded_static_ok(lg.shape[0], lg.shape[1], pt.slots)
                and self.matcher.padded_eligible(layers, lg.shape[1], pt.slots))

    def _forward_padded(self, outputs, pt):
        """criterion.py:163-207 on ``PaddedTargets``: padded cost blocks -> on-device Hungarian -> fused losses; the
        normaliser num_boxes (criterion.py:178-181) is the device scalar pt.num_boxes, which whoever loaded the batch
        has already averaged over the ranks.  Nothing here depends on the number of events per video."""
        main = {k: v for k, v in outputs.items() if k not in ('aux_outputs', 'enc_outputs')}
        layers = [main] + list(outputs.get('aux_outputs', []))
        self._tgt_cat = (pt.labels.view(-1), pt.boxes.view(-1, 2))
        self._gt_counts = pt.counts
        batched = self.matcher.match_layers_padded(layers, pt)
        outputs['matched_indices'] = batched[0]
        losses = self._fused_losses(layers, batched, pt.num_boxes)
        if 'aux_outputs' in outputs:
            return losses, batched[0], list(batched[1:])
        return losses, batched[0]

    def forward(self, outputs, targets):
        """criterion.py:163-207 -> (losses, last_indices[, aux_indices]).  targets: the reference's list of per-video
        dicts, or a gvl_amd.targets.PaddedTargets (fixed-shape form used by the captured steps)."""
        if hasattr(targets, "host_counts"):
            if self.padded_eligible(outputs, targets):
                return self._forward_padded(outputs, targets)
            if outputs['pred_logits'].is_cuda and torch.cuda.is_current_stream_capturing():
                # as_list() slices with the HOST counts of the batch at hand: captured, that layout would be replayed
                # against every later batch without an error
                raise RuntimeError("SetCriterion: PaddedTargets outside the padded path's domain while a hipGraph is "
                                   "being captured (the caller's padded_static_ok() check and padded_eligible() disagree)")
            targets = targets.as_list()
        main = {k: v for k, v in outputs.items() if k not in ('aux_outputs', 'enc_outputs')}
        aux_list = outputs.get('aux_outputs', [])
        dev = outputs['pred_logits'].device
        self._state = {}
        self._tgt_cat = (torch.cat([t_["labels"] for t_ in targets]).to(dev),
                         torch.cat([t_["boxes"] for t_ in targets]).to(dev))
        sizes_key = (tuple(len(t_["boxes"]) for t_ in targets), str(dev))
        cached = self._const_cache.get(sizes_key)
        if cached is None:        # constants of this batch layout live on the device once (no per-step host->device copy)
            if len(self._const_cache) >= 64:                    # bounded: one entry per batch layout seen
                self._const_cache.pop(next(iter(self._const_cache)))
            cached = self._const_cache[sizes_key] = torch.tensor(sizes_key[0], dtype=torch.long, device=dev)
        self._gt_counts = cached
        batched = None
        if hasattr(self.matcher, "match_layers_device") and dev.type == "cuda" and self.device_matching:
            # all layers x videos solved on the device in one launch: no device->host copy at all
            # (None when a problem exceeds the on-chip solver, e.g. Q = 300 with > 64 events in one video)
            batched = self.matcher.match_layers_device([main] + list(aux_list), targets)
        if batched is not None:
            last_indices = batched[0]
        elif hasattr(self.matcher, "match_layers"):
            # same matchings as one matcher call per layer, but one device->host copy for all layers
            batched = self.matcher.match_layers([main] + list(aux_list), targets)
            last_indices = batched[0]
        else:
            last_indices = self.matcher(main, targets)
        outputs['matched_indices'] = last_indices
        num_boxes = sum(len(t_["labels"]) for t_ in targets)
        # criterion.py:178-181.  Only while training: the reference never evaluates under a process group, and an
        # eval forward sharded by video must not synchronise the ranks (its losses are per-rank diagnostics)
        if isinstance(self.num_boxes_override, torch.Tensor):
            num_boxes = self.num_boxes_override.reshape(())             # device scalar: read by the kernels / ops
        elif self.num_boxes_override is not None:
            num_boxes = float(self.num_boxes_override)
        elif is_dist_avail_and_initialized() and torch.is_grad_enabled():
            nb = torch.as_tensor([num_boxes], dtype=torch.float, device=outputs['pred_logits'].device)
            dist.all_reduce(nb)
            num_boxes = torch.clamp(nb / get_world_size(), min=1).item()
        else:
            num_boxes = max(float(num_boxes), 1.0)                     # same value, no device round trip
        if (self.fused and batched is not None and hasattr(last_indices, "plan") and dev.type == "cuda"
                and set(self.losses) == {'labels', 'boxes', 'cardinality'} and outputs['pred_logits'].dtype == torch.float32
                and outputs['pred_logits'].shape[0] * outputs['pred_logits'].shape[1] <= 24576):
            losses = self._fused_losses([main] + list(aux_list), batched, num_boxes)
            if 'aux_outputs' in outputs:
                return losses, last_indices, list(batched[1:])
            return losses, last_indices
        losses = {}
        for loss in self.losses:
            losses.update(self.get_loss(loss, outputs, targets, last_indices, num_boxes))
        if 'aux_outputs' in outputs:
            aux_indices = []
            for i, aux in enumerate(outputs['aux_outputs']):
                indices = batched[i + 1] if batched is not None else self.matcher(aux, targets)
                aux_indices.append(indices)
                for loss in self.losses:
                    if loss == 'masks':
                        continue
                    kwargs = {'log': False} if loss == 'labels' else {}
                    l_dict = self.get_loss(loss, aux, targets, indices, num_boxes, **kwargs)
                    losses.update({k + f'_{i}': v for k, v in l_dict.items()})
            return losses, last_indices, aux_indices
        return losses, last_indices
