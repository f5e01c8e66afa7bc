"""Data-parallel train step for the PDVC hot path: one process per GPU, videos sharded by rank, bucketed gradient
all-reduce over RCCL (``torch.distributed`` backend "nccl" on ROCm) / gloo on CPU test rigs.

The reference trainer is single-GPU (train.py:286,598-599 never builds a process group); the only collective on its
path is ``all_reduce(num_boxes)`` inside the criterion (criterion.py:178-180), which gvl_amd.criterion keeps.  This
module adds what a data-parallel run of the reference loop (train.py:385-409) needs and nothing more:

  * every parameter's ``.grad`` is a view into one flat fp32 buffer cut into ~25 MB buckets in reverse registration
    order (gradients of the caption / box heads are produced first in backward, the base encoder's last);
  * a post-accumulate-grad hook counts a bucket's parameters and launches its all-reduce (async, on RCCL's stream) as
    soon as the bucket is complete, so the exchange of early buckets overlaps the rest of backward;
  * the 100.7 MB of gradients (25.17 M fp32 parameters) therefore travel as 4 large messages: on MI355X's full xGMI
    mesh (7 links x ~153 GB/s per GPU) large messages are what the ring/direct algorithms need to reach link rate;
  * after backward: wait, scale by 1/world, clip_grad_norm_, Adam step -- train.py:405-409.
"""
import torch
import torch.distributed as dist


def shard_batch(dt, rank, world):
    """rank r of W takes videos r::W (DistributedSampler-style); only used by tests / multi-GPU drivers."""
    B = dt["video_tensor"].shape[0]
    idx = list(range(rank, B, world))
    n_gt = [len(t_["boxes"]) for t_ in dt["video_target"]]
    starts = [sum(n_gt[:i]) for i in range(B)]
    cap_idx = [k for i in idx for k in range(starts[i], starts[i] + n_gt[i])]
    out = dict(dt)
    for k in ("video_tensor", "video_mask", "video_length", "gt_boxes_mask"):
        if k in dt:
            out[k] = dt[k][idx]
    out["video_target"] = [dt["video_target"][i] for i in idx]
    out["cap_raw"] = [dt["cap_raw"][i] for i in idx]
    for k in ("cap_tensor", "cap_mask"):
        if k in dt:
            out[k] = dt[k][cap_idx]
    return out


class GradBuckets:
    """Flat gradient storage + overlapped bucketed all-reduce."""

    def __init__(self, params, bucket_bytes=25 << 20, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        # reverse registration order ~ the order in which backward produces gradients
        self.buckets = []            # (start, end, n_params)
        self.bucket_of = {}
        cur_start, cur_n, off = 0, 0, 0
        for p in reversed(self.params):
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self.bucket_of[id(p)] = len(self.buckets)
            off += n
            cur_n += 1
            if (off - cur_start) * self.flat.element_size() >= bucket_bytes:
                self.buckets.append((cur_start, off, cur_n))
                cur_start, cur_n = off, 0
        if cur_n:
            self.buckets.append((cur_start, off, cur_n))
        self.ready = [0] * len(self.buckets)
        self.handles = []
        if self.world > 1:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)

    def _hook(self, p):
        b = self.bucket_of[id(p)]
        self.ready[b] += 1
        if self.ready[b] == self.buckets[b][2]:
            s, e, _ = self.buckets[b]
            self.handles.append(dist.all_reduce(self.flat[s:e], group=self.group, async_op=True))

    def zero(self):
        self.flat.zero_()
        self.ready = [0] * len(self.buckets)
        self.handles = []

    def finish(self):
        """wait for the in-flight buckets, reduce the ones whose parameters got no gradient this step, average."""
        if self.world == 1:
            return
        for b, (s, e, n) in enumerate(self.buckets):
            if self.ready[b] != n:                       # some parameter unused this step: reduce the bucket now
                self.handles.append(dist.all_reduce(self.flat[s:e], group=self.group, async_op=True))
        for h in self.handles:
            h.wait()
        self.flat.div_(self.world)


class TrainStep:
    """zero_grad -> forward -> weighted loss -> backward (+ overlapped all-reduce) -> clip -> Adam (train.py:385-409)"""

    def __init__(self, model, criterion, opt, world_size=1, process_group=None):
        self.model, self.criterion, self.opt = model, criterion, opt
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.buckets = GradBuckets(self.params, process_group=process_group)
        self.optimizer = torch.optim.Adam(self.params, lr=opt.lr, weight_decay=opt.weight_decay)
        self.world = world_size

    def __call__(self, dt):
        self.buckets.zero()                                                    # optimizer.zero_grad(), flat
        out, loss = self.model(dt, self.criterion, None, self.opt.transformer_input_type)
        wd = self.criterion.weight_dict
        final = sum(loss[k] * wd[k] for k in loss.keys() if k in wd)
        final.backward()
        self.buckets.finish()
        torch.nn.utils.clip_grad_norm_(self.params, self.opt.grad_clip)
        self.optimizer.step()
        return final.detach(), loss
