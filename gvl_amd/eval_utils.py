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


def gather_results(local_results, dst=0, group=None):
    """Evaluation sharded by video over the ranks: every rank holds the result records of
    ITS videos ({video_key: [prediction dicts]}, as batch_result_json returns them); rank `dst` receives the union and
    every other rank None.  The reference's helper for this is the pickle all_gather of misc/detr_utils/misc.py:183-223
    (size exchange + padded byte tensors); torch.distributed's object collectives do exactly that on either backend --
    RCCL ("nccl" on ROCm) moves the bytes through device memory, gloo through host memory.  One collective per call,
    after the last batch; the data path of the eval forward itself has none."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return dict(local_results)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = [None] * world if rank == dst else None
    dist.gather_object(dict(local_results), parts, dst=dst, group=group)
    if rank != dst:
        return None
    merged = {}
    for r, part in enumerate(parts):
        dup = set(part) & set(merged)
        if dup:
            raise ValueError(f"video keys evaluated on more than one rank (rank {r}): {sorted(dup)[:3]}")
        merged.update(part)
    return merged


def shard_indices(n_items, rank, world):
    """DistributedSampler-style shard without padding: rank r takes items r, r + W, ... (every video exactly once)"""
    return list(range(rank, n_items, world))
