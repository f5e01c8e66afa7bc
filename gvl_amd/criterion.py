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

from .matcher import box_cl_to_xy, box_iou, generalized_box_iou

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

    @staticmethod
    def _src_idx(indices):
        batch = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        return batch, torch.cat([src for (src, _) in indices])

    def loss_labels(self, outputs, targets, indices, num_boxes, log=True):
        indices, _ = indices
        logits = outputs['pred_logits']
        dev = logits.device
        b_idx, q_idx = self._src_idx(indices)
        tgt_o = torch.cat([t["labels"][J.to(t["labels"].device)] for t, (_, J) in zip(targets, indices)])
        classes = torch.full(logits.shape[:2], self.num_classes, dtype=torch.int64, device=dev)
        classes[b_idx.to(dev), q_idx.to(dev)] = tgt_o.to(dev)
        onehot = torch.zeros([logits.shape[0], logits.shape[1], logits.shape[2] + 1], dtype=logits.dtype, device=dev)
        onehot.scatter_(2, classes.unsqueeze(-1), 1)
        onehot = onehot[:, :, :-1]
        losses = {'loss_ce': sigmoid_focal_loss(logits, onehot, num_boxes, alpha=self.focal_alpha,
                                                gamma=self.focal_gamma) * logits.shape[1]}
        pred_count = outputs['pred_count']
        max_length = pred_count.shape[1] - 1
        counts = torch.tensor([min(len(t_['boxes']), max_length) for t_ in targets], device=dev, dtype=torch.long)
        cnt_onehot = torch.zeros_like(pred_count)
        cnt_onehot.scatter_(1, counts.unsqueeze(-1), 1)
        weight = self.counter_class_rate[:max_length + 1].to(dev)
        losses['loss_counter'] = cross_entropy_with_gaussian_mask(pred_count, cnt_onehot, self.opt, weight)
        return losses

    @torch.no_grad()
    def loss_cardinality(self, outputs, targets, indices, num_boxes):
        logits = outputs['pred_logits']
        tgt_lengths = torch.as_tensor([len(v["labels"]) for v in targets], device=logits.device)
        card_pred = (logits.argmax(-1) != logits.shape[-1] - 1).sum(1)
        return {'cardinality_error': F.l1_loss(card_pred.float(), tgt_lengths.float())}

    def loss_boxes(self, outputs, targets, indices, num_boxes):
        indices, _ = indices
        dev = outputs['pred_boxes'].device
        b_idx, q_idx = self._src_idx(indices)
        src = outputs['pred_boxes'][b_idx.to(dev), q_idx.to(dev)]
        tgt = torch.cat([t_['boxes'][i.to(t_['boxes'].device)] for t_, (_, i) in zip(targets, indices)], dim=0)
        losses = {'loss_bbox': F.l1_loss(src, tgt, reduction='none').sum() / num_boxes}
        giou = torch.diag(generalized_box_iou(box_cl_to_xy(src), box_cl_to_xy(tgt)))
        losses['loss_giou'] = (1 - giou).sum() / num_boxes
        self_iou = torch.triu(box_iou(box_cl_to_xy(src), box_cl_to_xy(src))[0], diagonal=1)
        sizes = [len(v[0]) for v in indices]
        total = 0
        for i, c in enumerate(self_iou.split(sizes, -1)):
            total = total + c.split(sizes, -2)[i].sum() / (0.5 * sizes[i] * (sizes[i] - 1))
        losses['loss_self_iou'] = total
        return losses

    def get_loss(self, loss, outputs, targets, indices, num_boxes, **kwargs):
        table = {'labels': self.loss_labels, 'cardinality': self.loss_cardinality, 'boxes': self.loss_boxes}
        assert loss in table, f'do you really want to compute {loss} loss?'
        return table[loss](outputs, targets, indices, num_boxes, **kwargs)

    def forward(self, outputs, targets):
        """criterion.py:163-207 -> (losses, last_indices[, aux_indices])"""
        main = {k: v for k, v in outputs.items() if k not in ('aux_outputs', 'enc_outputs')}
        last_indices = self.matcher(main, targets)
        outputs['matched_indices'] = last_indices
        num_boxes = sum(len(t_["labels"]) for t_ in targets)
        num_boxes = torch.as_tensor([num_boxes], dtype=torch.float, device=outputs['pred_logits'].device)
        if is_dist_avail_and_initialized():
            dist.all_reduce(num_boxes)
        num_boxes = torch.clamp(num_boxes / get_world_size(), min=1).item()
        losses = {}
        for loss in self.losses:
            losses.update(self.get_loss(loss, outputs, targets, last_indices, num_boxes))
        if 'aux_outputs' in outputs:
            aux_indices = []
            for i, aux in enumerate(outputs['aux_outputs']):
                indices = self.matcher(aux, targets)
                aux_indices.append(indices)
                for loss in self.losses:
                    if loss == 'masks':
                        continue
                    kwargs = {'log': False} if loss == 'labels' else {}
                    l_dict = self.get_loss(loss, aux, targets, indices, num_boxes, **kwargs)
                    losses.update({k + f'_{i}': v for k, v in l_dict.items()})
            return losses, last_indices, aux_indices
        return losses, last_indices
