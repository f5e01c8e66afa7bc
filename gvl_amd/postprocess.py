"""PostProcess -- mirror of pdvc/pdvc.py:932-1089 for the non-contrastive LSTM-DSA path (what eval_utils.py:213-216
calls on every eval batch): top-N_q events by confidence, boxes clipped to the video and scaled to seconds, captions
re-ordered accordingly and decoded with the dataset's translator, caption scores = sum of token log-probs.
Host glue; the only device work is a topk / gather."""
import torch
from torch import nn

from .matcher import box_cl_to_xy


class PostProcess(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt

    @torch.no_grad()
    def forward_grounding(self, outputs, target_sizes, targets):
        """pdvc.py:951-953: grounding needs the contrastive text branch, which this build does not carry."""
        return None, None

    @torch.no_grad()
    def forward(self, outputs, target_sizes, loader, model=None, tokenizer=None):
        out_logits, out_bbox = outputs['pred_logits'], outputs['pred_boxes']
        N, N_q, N_class = out_logits.shape
        assert len(out_logits) == len(target_sizes)
        prob = out_logits.sigmoid()
        scores, topk_indexes = torch.topk(prob.view(N, -1), N_q, dim=1)
        topk_boxes = topk_indexes // N_class
        labels = topk_indexes % N_class
        raw_boxes = box_cl_to_xy(out_bbox)
        boxes = raw_boxes.clamp(min=0, max=1)
        boxes = torch.gather(boxes, 1, topk_boxes.unsqueeze(-1).repeat(1, 1, 2))
        boxes = boxes * torch.stack([target_sizes, target_sizes], dim=1)[:, None, :]
        seq = outputs['seq']
        cap_prob = outputs['caption_probs']['cap_prob_eval']
        eseq_lens = outputs['pred_count'].argmax(dim=-1).clamp(min=1)
        bs, num_queries = boxes.shape[:2]
        if len(seq):
            mask = (seq > 0).float()
            cap_scores = (mask * cap_prob).sum(2).cpu().numpy().astype('float')
            order = topk_boxes.cpu().tolist()
            seq_np = seq.detach().cpu().numpy().astype('int')
            caps = [[loader.dataset.translator.rtranslate(s) for s in s_vid] for s_vid in seq_np]
            caps = [[caps[b][idx] for idx in row] for b, row in enumerate(order)]
            cap_scores = [[cap_scores[b, idx] for idx in row] for b, row in enumerate(order)]
        else:
            cap_scores = [[-1e5] * num_queries] * bs
            caps = [[''] * num_queries] * bs
        cl_scores = [[0.0] * num_queries] * bs
        return [{'scores': s, 'labels': l, 'boxes': b, 'raw_boxes': b, 'captions': c, 'caption_scores': cs,
                 'cl_scores': cls, 'query_id': qid, 'vid_duration': ts, 'pred_seq_len': sl, 'raw_idx': idx}
                for s, l, b, c, cs, cls, qid, ts, sl, idx in
                zip(scores, labels, boxes, caps, cap_scores, cl_scores, topk_boxes, target_sizes, eseq_lens,
                    topk_indexes)]
