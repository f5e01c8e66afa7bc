"""PostProcess -- the result side of the eval loop, mirror of pdvc/pdvc.py:932-1089 for the non-contrastive LSTM-DSA
path (what eval_utils.py:213-216 calls on every eval batch).  Per video: the N_q most confident (query, class) pairs,
their segments clipped to the video and scaled to seconds, their captions decoded with the dataset's translator and
scored by the sum of the token log-probabilities.  Host glue; the only device work is a top-k and a gather.
The record keys are the reference's wire format (eval_utils.py:216-239 reads them)."""
import torch
from torch import nn

from .matcher import box_cl_to_xy

RECORD_KEYS = ('scores', 'labels', 'boxes', 'raw_boxes', 'captions', 'caption_scores', 'cl_scores', 'query_id',
               'vid_duration', 'pred_seq_len', 'raw_idx')


def rank_events(pred_logits):
    """confidence ranking over all (query, class) pairs of a video (pdvc.py:1019-1023)
    -> (scores (N, N_q) descending, flat pair index, query id, class id)"""
    n_vid, n_query, n_class = pred_logits.shape
    scores, flat = torch.topk(pred_logits.sigmoid().reshape(n_vid, n_query * n_class), n_query, dim=1)
    return scores, flat, flat // n_class, flat % n_class


def segments_in_seconds(pred_boxes, query_id, durations):
    """(centre, length) in [0, 1] -> [start, end] seconds of the ranked queries, clipped to the video (pdvc.py:1024-1031)"""
    spans = box_cl_to_xy(pred_boxes).clamp(0, 1)
    picked = spans.gather(1, query_id[..., None].expand(-1, -1, 2))
    return picked * durations[:, None, None]


def decode_captions(seq, log_probs, query_id, translator):
    """token ids + per-token log-probs of every query -> (sentences, scores) re-ordered by the ranking (pdvc.py:1046-1066)"""
    n_vid, n_query = query_id.shape
    if not len(seq):                                         # captioning disabled / every caption empty
        return [[''] * n_query] * n_vid, [[-1e5] * n_query] * n_vid
    total = ((seq > 0).float() * log_probs).sum(2).cpu().numpy().astype('float')
    tokens = seq.detach().cpu().numpy().astype('int')
    ranking = query_id.cpu().tolist()
    sentences = [[translator.rtranslate(tokens[v][q]) for q in ranking[v]] for v in range(n_vid)]
    scores = [[total[v, q] for q in ranking[v]] for v in range(n_vid)]
    return sentences, scores


class PostProcess(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt

    @torch.no_grad()
    def forward_grounding(self, outputs, target_sizes, targets):
        """pdvc.py:949-951: without the contrastive text branch the reference returns (None, None) here too."""
        return None, None

    @torch.no_grad()
    def forward(self, outputs, target_sizes, loader, model=None, tokenizer=None):
        logits = outputs['pred_logits']
        assert len(logits) == len(target_sizes)
        scores, flat, query_id, class_id = rank_events(logits)
        segments = segments_in_seconds(outputs['pred_boxes'], query_id, target_sizes)
        sentences, sentence_scores = decode_captions(outputs['seq'], outputs['caption_probs']['cap_prob_eval'], query_id,
                                                     loader.dataset.translator)
        n_events = outputs['pred_count'].argmax(dim=-1).clamp(min=1)          # predicted number of events, at least one
        no_cl = [0.0] * query_id.shape[1]                                     # contrastive scores: branch not built
        columns = (scores, class_id, segments, segments, sentences, sentence_scores, [no_cl] * len(logits), query_id,
                   target_sizes, n_events, flat)
        return [dict(zip(RECORD_KEYS, row)) for row in zip(*columns)]
