"""Batch assembly in front of the path -- mirror of ``collate_fn`` (video_dataset.py:16-106): a list of per-video samples
``(features (T_i, C) float32 ndarray, gt featstamps, labels, captions (list of int arrays), raw timestamps (list of
[start, end] seconds), duration, raw caption strings, key)`` -> the flat ``dt`` dict PDVC.forward consumes (keys
``video_tensor, video_length, video_mask, video_key, video_target, gt_featstamps, gt_timestamp, gt_gather_idx, gt_boxes,
gt_boxes_mask, cap_tensor, cap_length, cap_mask, cap_raw``).  Host-side data format only; the
dataset classes that read feature files / annotation JSON are out of scope."""
from itertools import chain

import numpy as np
import torch


def _boxes(raw_timestamps, duration):
    """(start, end) seconds -> normalised (centre, length), video_dataset.py:60-62,76-78"""
    return torch.tensor([[(ts[1] + ts[0]) / (2 * duration), (ts[1] - ts[0]) / duration] for ts in raw_timestamps]).float()


def collate_fn(batch):
    feats, featstamps, labels, captions, raw_ts, durations, raw_caps, keys = zip(*batch)
    B, C = len(batch), feats[0].shape[1]
    t_max = max(x.shape[0] for x in feats)
    n_cap = [len(c) for c in captions]
    cap_max = max(chain(*[[len(c) for c in cs] for cs in captions]))
    video_tensor = torch.zeros(B, t_max, C, dtype=torch.float32)
    video_length = torch.zeros(B, 3, dtype=torch.float32)            # (frames, duration in seconds, number of events)
    video_mask = torch.zeros(B, t_max, dtype=torch.bool)
    cap_tensor = torch.zeros(sum(n_cap), cap_max, dtype=torch.long)
    cap_length = torch.zeros(sum(n_cap), dtype=torch.long)
    cap_mask = torch.zeros(sum(n_cap), cap_max, dtype=torch.bool)
    gather_idx = torch.zeros(sum(n_cap), dtype=torch.long)
    gt_boxes = torch.zeros(B, max(n_cap), 2)
    row = 0
    for i in range(B):
        T, n = feats[i].shape[0], len(featstamps[i])
        video_tensor[i, :T] = torch.from_numpy(np.ascontiguousarray(feats[i]))
        video_length[i] = torch.tensor([float(T), float(durations[i]), float(n)])
        video_mask[i, :T] = True
        gather_idx[row:row + n] = i
        gt_boxes[i, :n] = _boxes(raw_ts[i], durations[i])
        for k, cap in enumerate(captions[i]):
            cap_length[row + k] = len(cap)
            cap_tensor[row + k, :len(cap)] = torch.from_numpy(np.asarray(cap))
            cap_mask[row + k, :len(cap)] = True
        row += n
    target = [{'boxes': _boxes(raw_ts[i], durations[i]), 'labels': torch.tensor(labels[i]).long(), 'masks': None,
               'image_id': keys[i]} for i in range(B)]
    return {
        "video_tensor": video_tensor, "video_length": video_length, "video_mask": video_mask, "video_key": list(keys),
        "video_target": target,
        "gt_featstamps": list(chain(*featstamps)), "gt_timestamp": list(raw_ts), "gt_gather_idx": gather_idx,
        "gt_boxes": gt_boxes, "gt_boxes_mask": (gt_boxes != 0).sum(2) > 0,
        "cap_tensor": cap_tensor, "cap_length": cap_length, "cap_mask": cap_mask, "cap_raw": list(raw_caps),
    }
