"""Fixed-shape, device-resident form of a batch's ground truth -- what lets ONE captured hipGraph serve every batch.

The reference hands the criterion / matcher / captioner a Python list of per-video dicts (``dt['video_target']``,
train.py:396-398, eval_utils.py:196-198) and caption tensors whose shapes follow the batch: (sum nGT, longest caption)
(video_dataset.py:68-88).  Every one of those sizes differs from batch to batch, so a step captured on one batch's
layout could not be replayed on the next.  ``PaddedTargets`` lays the same data out with NO data-dependent shape:

    boxes   (B, slots, 2)  float32     labels (B, slots) int64      counts (B,) int64   <- how many slots are real
    cap_tensor (B, slots, cap_len) int64    cap_mask (B, slots, cap_len) float32          (training only)
    num_boxes (1,) float32   the normaliser of criterion.py:178-181 (already averaged over the ranks)

Target k of video v sits in slot (v, k); its caption in the same slot.  The matcher's problem descriptors, the
criterion kernels and the caption gather read ``counts`` on the device (include/gvl_msda.h: gvl_match_cost_padded_f32,
video_pair_count / num_boxes_dev of gvl_set_criterion_*), so a replay only needs ``load()`` -- a handful of small
copies into these static buffers.  ``slots`` and ``cap_len`` are capacities chosen by the caller
(gvl_amd.parallel: grow-only, rounded up to a power of two / multiple of 4).
"""
import torch


def round_up_pow2(n, floor=4):
    v = floor
    while v < n:
        v *= 2
    return v


def needed_capacity(dt):
    """(largest number of events of one video, caption tensor width) of a reference-format batch; host-side only."""
    n_gt = max([len(t_["boxes"]) for t_ in dt["video_target"]] + [0])
    cap = dt.get("cap_tensor")
    cap_len = int(cap.shape[-1]) if isinstance(cap, torch.Tensor) and cap.dim() == 2 else 0
    return n_gt, cap_len


def total_events(dt):
    return sum(len(t_["boxes"]) for t_ in dt["video_target"])


class PaddedTargets:
    def __init__(self, B, slots, cap_len, device, pair_rows=0):
        """pair_rows: capacity of the compact caption-row set of the train step (>= the number of events of a whole
        batch; 0 = B * slots, i.e. no compaction)"""
        self.B, self.slots, self.cap_len, self.device = B, slots, cap_len, torch.device(device)
        self.pair_rows = int(pair_rows or 0)
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=self.device)      # noqa: E731
        self.boxes, self.labels = z(B, slots, 2), z(B, slots, dtype=torch.int64)
        self.counts = z(B, dtype=torch.int64)
        self.num_boxes = torch.ones(1, dtype=torch.float32, device=self.device)
        self.cap_tensor = z(B, slots, cap_len, dtype=torch.int64) if cap_len else None
        self.cap_mask = z(B, slots, cap_len) if cap_len else None
        self.host_counts = [0] * B
        # ring of pinned staging buffers for the one host->device copy of load(): the host runs ahead of the GPU when
        # steps are graph replays, so a buffer is reused only after the copy that last read it has completed
        cuda = self.device.type == "cuda"
        self._ring = [(torch.empty(B * slots + B, dtype=torch.int64).pin_memory() if cuda
                       else torch.empty(B * slots + B, dtype=torch.int64), torch.cuda.Event() if cuda else None)
                      for _ in range(4)]
        self._turn = 0

    def fits(self, dt):
        n_gt, cap_len = needed_capacity(dt)
        return (len(dt["video_target"]) == self.B and n_gt <= self.slots
                and (self.cap_len == 0 or cap_len <= self.cap_len))

    @torch.no_grad()
    def load(self, dt, num_boxes=None):
        """refresh the static buffers from a reference-format batch.  num_boxes: the cross-rank mean of the target
        count when a process group exists (criterion.py:178-181); default = this batch's own count, floored at 1."""
        targets = dt["video_target"]
        ns = [len(t_["boxes"]) for t_ in targets]
        if len(ns) != self.B or max(ns + [0]) > self.slots:
            raise ValueError(f"PaddedTargets(B={self.B}, slots={self.slots}) cannot hold a batch with event counts {ns}")
        total = sum(ns)
        if self.pair_rows and total > self.pair_rows:
            raise ValueError(f"PaddedTargets(pair_rows={self.pair_rows}) cannot hold a batch with {total} events")
        self.host_counts = ns
        nb = float(max(total, 1)) if num_boxes is None else float(num_boxes)
        pos = [v * self.slots + k for v, n in enumerate(ns) for k in range(n)]
        st, ev = self._ring[self._turn]
        self._turn = (self._turn + 1) % len(self._ring)
        if ev is not None:
            ev.synchronize()
        st[:total + self.B] = torch.tensor(pos + ns, dtype=torch.int64)
        packed = st[:total + self.B].to(self.device, non_blocking=True)
        if ev is not None:
            ev.record()
        idx = packed[:total]
        self.counts.copy_(packed[total:])
        self.num_boxes.fill_(nb)
        self.boxes.zero_()
        self.labels.zero_()
        if total:
            boxes = torch.cat([t_["boxes"] for t_ in targets]).to(self.device, torch.float32)
            labels = torch.cat([t_["labels"] for t_ in targets]).to(self.device, torch.int64)
            self.boxes.view(-1, 2).index_copy_(0, idx, boxes)
            self.labels.view(-1).index_copy_(0, idx, labels)
        if self.cap_tensor is not None:
            self.cap_tensor.zero_()
            self.cap_mask.zero_()
            cap = dt.get("cap_tensor")
            if total and isinstance(cap, torch.Tensor) and cap.dim() == 2:
                if cap.shape[0] != total or cap.shape[1] > self.cap_len:
                    raise ValueError(f"cap_tensor {tuple(cap.shape)} does not match {total} targets / cap_len {self.cap_len}")
                w = cap.shape[1]
                self.cap_tensor.view(-1, self.cap_len)[:, :w].index_copy_(0, idx, cap.to(self.device, torch.int64))
                self.cap_mask.view(-1, self.cap_len)[:, :w].index_copy_(0, idx, dt["cap_mask"].to(self.device,
                                                                                                 torch.float32))
        return self

    def as_list(self):
        """back to the reference's list-of-dicts form (host-known counts; used by paths that have no padded variant)"""
        return [{"boxes": self.boxes[v, :n], "labels": self.labels[v, :n]} for v, n in enumerate(self.host_counts)]
