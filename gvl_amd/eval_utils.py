"""Result wire format of the evaluation loop -- mirror of the dict construction in eval_utils.py:216-239 and of
``save_dvc_json`` (eval_utils.py:136-141): PostProcess results -> ``{video_key: [ {timestamp, raw_box, label,
proposal_score, sentence, sentence_score, cl_score, query_id, vid_duration, pred_event_count}, ... ]}``.

Host-side, data-format only; the reference's grounding / TAL / plotting branches and the
Java / pycocoevalcap scorers that consume the file are out of scope."""
import json


def batch_result_json(results, video_keys, score_threshold=0):
    """eval_utils.py:216-239 for one batch: keeps predictions whose proposal score exceeds the threshold and whose raw
    box is not all-zero."""
    batch_json = {}
    for idx, video_name in enumerate(video_keys):
        r = results[idx]
        segment = r['boxes'].cpu().numpy()
        raw_boxes = r['raw_boxes'].cpu().numpy()
        raw_boxes_mask = raw_boxes.sum(1) != 0
        batch_json[video_name] = [
            {
                "timestamp": segment[pid].tolist(),
                "raw_box": raw_boxes[pid].tolist(),
                "label": r['labels'][pid].item(),
                "proposal_score": r['scores'][pid].item(),
                "sentence": r['captions'][pid],
                "sentence_score": r['caption_scores'][pid],
                "cl_score": r['cl_scores'][pid],
                'query_id': r['query_id'][pid].item(),
                'vid_duration': r['vid_duration'].item(),
                'pred_event_count': r['pred_seq_len'].item(),
            }
            for pid in range(len(segment)) if r['scores'][pid].item() > score_threshold and raw_boxes_mask[pid]]
    return batch_json


def new_result_file():
    """skeleton written by the evaluation loop (eval_utils.py:174-176)"""
    return {'results': {}, 'version': 'VERSION 1.0', 'external_data': {'used:': True, 'details': None}}


def save_dvc_json(out_json, path, verbose=False):
    """eval_utils.py:136-141"""
    with open(path, 'w') as f:
        if verbose:
            out_json['valid_video_num'] = len(out_json['results'])
            n = [len(v) for v in out_json['results'].values()]
            out_json['avg_proposal_num'] = float(sum(n)) / len(n) if n else float('nan')
        json.dump(out_json, f)
