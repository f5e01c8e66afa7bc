"""Batch assembly in front of the path -- mirror of ``collate_fn`` (video_dataset.py:16-106): a list of per-video samples
``(features (T_i, C) float32 ndarray, gt featstamps, labels, captions (list of int arrays), raw timestamps (list of
[start, end] seconds), duration, raw caption strings, key)`` -> the flat ``dt`` dict PDVC.forward consumes (keys
``video_tensor, video_length, video_mask, video_key, video_target, gt_featstamps, gt_timestamp, gt_gather_idx, gt_boxes,
gt_boxes_mask, cap_tensor, cap_length, cap_mask, cap_raw``), and of the dataset that produces those samples
(``PropSeqDataset``: annotation JSON + one feature file per video, nearest temporal rescale, zeros for missing files).
Host-side data formats only."""
import json
import os
import pickle
from collections import defaultdict
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


# ---------------------------------------------------------------------------------------------------------------
# The sample side of f4: what produces the tuples collate_fn consumes.  Mirror of PropSeqDataset.__getitem__
# (video_dataset.py:264-281) with load_feats (:209-247), get_feats / read_file (:307-384), resizeFeature (:386-397),
# Translator (:109-137) and process_time_step (:193-200).  Pure host code (numpy / json), no GPU on this side.
# ---------------------------------------------------------------------------------------------------------------
# feature type -> (dimension, file name from the video key, normalisation mean, variance)   video_dataset.py:309-357
_FEATURE_TYPES = {
    "c3d": (500, lambda k: k[0:13] + ".npy", -0.001915027447565527, 1.9239444588254049),
    "c3d4096": (4096, lambda k: k + ".npy", 0, 0),
    "resnet": (2048, lambda k: k[2:13] + "_resnet.npy", 0.41634243404998694, 0.2569392081183313),
    "bn": (1024, lambda k: k[2:13] + "_bn.npy", 0.8945046635916155, 3.6579982046018844),
    "tsn_100": (400, lambda k: k[0:13] + ".csv", 0, 0),
    "i3d_rgb": (1024, lambda k: k[:13] + "_rgb.npy", 0, 0),
    "i3d_flow": (1024, lambda k: k[:13] + "_flow.npy", 0, 0),
    "tsp": (512, lambda k: k[0:13] + ".npy", 0, 0),
    "swin": (1024, lambda k: k[0:13] + ".npy", 0, 0),
    "vggish": (128, lambda k: k[0:13] + ".npy", 0, 0),
    "clip_pkl": (768, lambda k: k[0:11] + ".pkl", 0, 0),
    "clip": (768, lambda k: k[0:13] + ".npy", 0, 0),
}
MISSING_FEATURE_FRAMES = 100          # a video without a feature file becomes 100 all-zero frames (video_dataset.py:319-322)


def read_features(key, vf_type, folder, data_norm=False):
    """-> (features (T, dim) ndarray, is_padding).  A missing file yields zeros((100, dim)) and is_padding = True."""
    if vf_type not in _FEATURE_TYPES:
        raise AssertionError('feature type error: {}'.format(vf_type))
    dim, name, mean, var = _FEATURE_TYPES[vf_type]
    path = os.path.join(folder, name(key))
    missing = not os.path.exists(path)
    if missing:
        print('{} not exists, use zero padding. '.format(path))
        feats = np.zeros((MISSING_FEATURE_FRAMES, dim))
    elif path.endswith(".npy"):
        feats = np.load(path)
    elif path.endswith(".csv"):
        import pandas as pd
        feats = pd.read_csv(path).values
    elif path.endswith(".pkl"):
        with open(path, "rb") as f:
            feats = pickle.load(f)
    else:
        raise NotImplementedError
    if data_norm:
        feats = (feats - mean) / np.sqrt(var)
    assert feats.ndim == 2 and feats.shape[1] == dim, 'load {} error, got shape {}'.format(path, feats.shape)
    return feats, missing


def resize_feature(x, new_size):
    """Nearest-neighbour temporal rescale to `new_size` frames: frame i of the result is the input frame nearest to
    i * (T - 1) / (new_size - 1), halves rounding DOWN (scipy interp1d(kind='nearest'), video_dataset.py:386-397);
    a one-frame input is repeated."""
    T = len(x)
    if T == 1:
        return np.stack([np.reshape(x, [-1])] * new_size)
    pos = np.array([i * float(T - 1) / (new_size - 1) for i in range(new_size)])
    idx = np.clip(np.ceil(pos - 0.5), 0, T - 1).astype(np.intp)
    return np.asarray(x)[idx]


def feature_stamps(duration, timestamps, n_frames):
    """seconds -> frame indices, clipped to the clip (video_dataset.py:193-200)"""
    stamps = np.array(n_frames) * np.array(timestamps) / np.array(duration)
    stamps = np.minimum(stamps, np.array(n_frames) - 1).astype('int')
    return np.maximum(stamps, 0).astype('int').tolist()


class Translator:
    """word <-> index maps of the vocabulary JSON (video_dataset.py:109-137); unknown words map to vocab_size."""
    _STRIP = ['!', '@', '%', '^', '*', '|', '#', '[', ']', '$', ',', ':', '!', '_', ';', '.', '?', '"', '\\n', '\\', '.']

    def __init__(self, vocab_json, vocab_size):
        self.vocab_size = vocab_size
        with open(vocab_json) as f:
            self.vocab = json.load(f)
        assert self.vocab_size == len(self.vocab['word_to_ix'].keys())
        self.vocab['word_to_ix'] = defaultdict(lambda: self.vocab_size, self.vocab['word_to_ix'])
        self.vocab['ix_to_word'] = defaultdict(lambda: self.vocab_size, self.vocab['ix_to_word'])

    def translate(self, sentence, max_len):
        for tok in self._STRIP:
            sentence = sentence.replace(tok, ' ')
        words = sentence.replace('.', ' . ').replace(',', ' , ').lower().split()
        return np.array([0] + [self.vocab['word_to_ix'][w] for w in words][:max_len - 2] + [0])

    def rtranslate(self, ids):
        ids = list(ids)
        if 0 in ids:
            ids = ids[:ids.index(0)]
        return ' '.join(self.vocab['ix_to_word'][str(i)] for i in ids) + '.' if ids else ''


class PropSeqDataset(torch.utils.data.Dataset):
    """The reference's training / evaluation dataset (video_dataset.py:149-281): annotation JSON {key: {duration,
    timestamps, sentences}} + one feature file per video -> the per-video tuples `collate_fn` consumes.  Features are
    rescaled to opt.frame_embedding_num frames (data_rescale) or sub-sampled (feature_sample_rate); at most
    opt.gt_proposal_sample_num events are kept, chosen with numpy's global RNG exactly as the reference does."""

    def __init__(self, anno_file, feature_folder, translator_json, is_training, proposal_type, opt):
        super().__init__()
        self.opt = opt
        self.translator = Translator(translator_json, opt.vocab_size)
        with open(anno_file) as f:
            self.anno = json.load(f)
        self.keys = list(self.anno.keys())
        for path in getattr(opt, "invalid_video_json", []) or []:
            with open(path) as f:
                bad = json.load(f)
            self.keys = [k for k in self.keys if k[:13] not in bad]
        self.feature_folder, self.is_training, self.proposal_type = feature_folder, is_training, proposal_type
        self.split_anno = vars(opt).get('train_with_split_anno', False)

    def __len__(self):
        return len(self.keys)

    def load_feats(self, key):
        opt = self.opt
        types, folders = opt.visual_feature_type, self.feature_folder
        rescale = bool(vars(opt).get("data_rescale", 1))
        if isinstance(types, list):
            assert isinstance(folders, list) and len(types) == len(folders)
            parts, all_missing = [], True
            for vf_type, folder in zip(types, folders):
                feats, missing = read_features(key, vf_type, folder)
                all_missing &= missing
                if rescale:
                    if feats.shape[0] != opt.frame_embedding_num:
                        feats = resize_feature(feats, opt.frame_embedding_num)
                else:
                    feats = feats[::vars(opt).get("feature_sample_rate", 1)]
                parts.append(feats)
            if all_missing:
                print('all feature files of video {} do not exist'.format(key))
            out = np.concatenate(parts, axis=-1)
        else:
            out, _ = read_features(key, types, folders, data_norm=vars(opt).get("data_norm", 0))
            if rescale:
                out = resize_feature(out, opt.frame_embedding_num)
        assert out.shape[1] == opt.feature_dim, 'wrong value of feature_dim'
        return out

    def __getitem__(self, idx):
        key = str(self.keys[idx])
        a = self.anno[key]
        duration, captions, stamps = a['duration'], a['sentences'], a['timestamps']
        labels = self.anno.get('action_labels', [0] * len(stamps))
        feats = self.load_feats(key[3:] if self.split_anno else key)
        keep = min(len(stamps), self.opt.gt_proposal_sample_num)
        chosen = set(np.random.choice(list(range(len(stamps))), keep, replace=False).tolist())
        captions = [c for i, c in enumerate(captions) if i in chosen]
        stamps = [s for i, s in enumerate(stamps) if i in chosen]
        labels = [x for i, x in enumerate(labels) if i in chosen]
        caption_ids = [np.array(self.translator.translate(s, self.opt.max_caption_len)) for s in captions]
        return (feats, feature_stamps(duration, stamps, feats.shape[0]), labels, caption_ids, stamps, duration, captions,
                key)
