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
from collections import OrderedDict

import os

import torch
import torch.distributed as dist

from .linear import deferred_wgrads
from .optim import bump_versions
from .targets import PaddedTargets, needed_capacity, round_up_pow2, total_events


def shard_batch(dt, rank, world):
    """rank r of W takes videos r::W (DistributedSampler-style); only used by tests / multi-GPU drivers.  B need not be a
    multiple of W: the leading B % W ranks hold one video more; a rank without any video (B < W) gets None and skips the
    forward (an eval forward has no collective; the result gather, gvl_amd.eval_utils.gather_results, takes an empty
    dict from it)."""
    B = dt["video_tensor"].shape[0]
    idx = list(range(rank, B, world))
    if not idx:
        return None
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

    def __init__(self, params, bucket_bytes=25 << 20, process_group=None, flat=None, overlap=True, first=None):
        """flat=None: flat buffer exactly when there is more than one process.  overlap=False: no autograd hooks --
        the buckets are exchanged by an explicit ``exchange()`` after backward (what a captured forward/backward
        needs: collectives stay outside the hipGraph).
        first: parameters whose gradients are complete EARLY (a two-stage backward: everything downstream of a cut);
        they are laid out first, with a bucket boundary behind them, so that ``exchange_begin(0)`` can put them on the
        wire while the second stage still runs (``self.n_first`` buckets)."""
        self.params = [p for p in params if p.requires_grad]
        first_ids = {id(p) for p in (first or [])}
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # One process, nothing to exchange: leave .grad = None so that autograd hands every gradient over without the
        # per-parameter accumulate kernel into a pre-existing buffer (133 launches per step; the step is launch-bound).
        self.flat = None
        self.buckets, self.bucket_of, self.ready, self.handles = [], {}, [], []
        self.next_bucket, self.seen = 0, set()
        self.n_first = 0
        self.overlap = overlap
        if not (self.world > 1 if flat is None else flat):
            return
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        # reverse registration order ~ the order in which backward produces gradients
        self.buckets = []            # (start, end, n_params)
        self.bucket_of = {}
        cur_start, cur_n, off = 0, 0, 0
        rev = list(reversed(self.params))
        order = [p for p in rev if id(p) in first_ids] + [p for p in rev if id(p) not in first_ids]
        n_lead = sum(1 for p in rev if id(p) in first_ids)
        self.n_first = 0
        for k, p in enumerate(order):
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self.bucket_of[id(p)] = len(self.buckets)
            off += n
            cur_n += 1
            if (off - cur_start) * self.flat.element_size() >= bucket_bytes or (n_lead and k == n_lead - 1):
                self.buckets.append((cur_start, off, cur_n))
                cur_start, cur_n = off, 0
                if n_lead and k == n_lead - 1:
                    self.n_first = len(self.buckets)
        if cur_n:
            self.buckets.append((cur_start, off, cur_n))
        self.ready = [0] * len(self.buckets)
        self.handles = []
        if self.world > 1 and overlap:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)

    def _launch_ready(self, everything=False):
        """issue the all-reduces in BUCKET ORDER: bucket b only after buckets 0..b-1.  Ranks whose backward completes the
        buckets in different orders (or leaves some incomplete: parameters without a gradient this step) still post the
        same sequence of collectives, so RCCL never pairs buckets of different sizes."""
        while self.next_bucket < len(self.buckets):
            s, e, n = self.buckets[self.next_bucket]
            if not everything and self.ready[self.next_bucket] != n:
                return
            self.handles.append(dist.all_reduce(self.flat[s:e], group=self.group, async_op=True))
            self.next_bucket += 1

    def _hook(self, p):
        self.seen.add(id(p))
        self.ready[self.bucket_of[id(p)]] += 1
        self._launch_ready()

    def zero(self):
        if self.flat is None:
            for p in self.params:
                p.grad = None
            return
        self.flat.zero_()
        self.ready = [0] * len(self.buckets)
        self.handles = []
        self.next_bucket, self.seen = 0, set()

    def exchange(self):
        """non-overlapped variant: all-reduce every bucket now (on the current stream's ordering) and average"""
        if self.world == 1 or self.flat is None:
            return
        handles = [dist.all_reduce(self.flat[s:e], group=self.group, async_op=True) for s, e, _ in self.buckets]
        for h in handles:
            h.wait()
        self.flat.div_(self.world)

    def exchange_begin(self, stage):
        """two-stage form: stage 0 = the `first` buckets (complete after the first backward stage), stage 1 = the rest.
        The all-reduces are posted asynchronously (RCCL's stream waits for the work already queued on the current stream,
        later work on the current stream overlaps them); ``exchange_end()`` waits for all of them and averages."""
        if self.world == 1 or self.flat is None:
            return
        part = self.buckets[:self.n_first] if stage == 0 else self.buckets[self.n_first:]
        self.handles += [dist.all_reduce(self.flat[s:e], group=self.group, async_op=True) for s, e, _ in part]

    def exchange_end(self):
        if self.world == 1 or self.flat is None:
            return
        for h in self.handles:
            h.wait()
        self.handles = []
        self.flat.div_(self.world)

    def finish(self):
        """wait for the in-flight buckets, reduce the ones whose parameters got no gradient this step, average."""
        if self.world == 1 or self.flat is None:
            return
        if not self.overlap:
            return self.exchange()
        self._launch_ready(everything=True)              # buckets with a parameter unused this step, in index order
        for h in self.handles:
            h.wait()
        self.flat.div_(self.world)

    def unused_params(self):
        """parameters that received no gradient ON ANY RANK in the step just finished (overlap mode).  The single-GPU
        loop leaves their .grad = None and Adam skips them (no weight decay, no moment update); with the flat buffer they
        hold zeros, so the step hides them from the optimizer.  The set is agreed across the ranks with one small MAX
        all-reduce of a per-parameter flag vector: a parameter some rank did use keeps its (averaged) gradient everywhere,
        so the replicas cannot drift apart on rank-dependent usage."""
        if self.flat is None or not self.overlap or self.world == 1:
            return []
        used = torch.tensor([1 if id(p) in self.seen else 0 for p in self.params], dtype=torch.int32,
                            device=self.flat.device)
        dist.all_reduce(used, op=dist.ReduceOp.MAX, group=self.group)
        used = used.tolist()
        return [p for p, u in zip(self.params, used) if not u]

    def agree_unused(self, unused_local, has_events):
        """ONE agreement, outside any capture, on the parameters the step's autograd graph can never reach: a
        parameter belongs to the set when every rank whose batch HAD events found it unused (`unused_local`: what
        probe_unused returned on this rank's batch); a rank without events abstains -- its probe also misses the captioner,
        which the other ranks do reach.  -> the list (identical on every rank), or None when no rank had events (nothing can be
        concluded from this step: ask again on a later one)."""
        ids = {id(p) for p in unused_local}
        if self.world == 1 or not dist.is_initialized():
            return [p for p in self.params if id(p) in ids] if has_events else None
        dev = self.flat.device if self.flat is not None else torch.device("cpu")
        member = torch.tensor([1 if (not has_events or id(p) in ids) else 0 for p in self.params], dtype=torch.int32, device=dev)
        voters = torch.tensor([1 if has_events else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(member, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(voters, op=dist.ReduceOp.MAX, group=self.group)
        if int(voters.item()) == 0:
            return None
        return [p for p, m in zip(self.params, member.tolist()) if m]

    def probe_unused(self, run):
        """no-hook (captured) mode: run one eager step `run()` with temporary post-accumulate hooks and return the
        parameters autograd never delivered a gradient to.  With the flat buffer installed their .grad is a zero view,
        not None, so a captured clip + Adam would decay them (weight_decay) and advance their moments where the
        single-GPU step and the reference (train.py:405-409) skip them."""
        seen = set()
        handles = [p.register_post_accumulate_grad_hook(lambda p_: seen.add(id(p_))) for p in self.params]
        try:
            run()
        finally:
            for h in handles:
                h.remove()
        return [p for p in self.params if id(p) not in seen]


class TrainStep:
    """zero_grad -> forward -> weighted loss -> backward (+ overlapped all-reduce) -> clip -> Adam (train.py:385-409)"""

    def __init__(self, model, criterion, opt, world_size=1, process_group=None, capturable=False, flat=None,
                 overlap=True, autocast_dtype=None, first=None):
        """autocast_dtype (e.g. torch.bfloat16): forward + losses under torch.autocast -- bf16 GEMMs and bf16-storage
        deformable attention, fp32 master weights / Adam, fp32 islands for the teacher-forced captioner loop and the
        criterion kernels."""
        self.model, self.criterion, self.opt = model, criterion, opt
        self.autocast_dtype = autocast_dtype
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.buckets = GradBuckets(self.params, process_group=process_group, flat=flat, overlap=overlap, first=first)
        # fused=True: one multi-tensor kernel per ~100 parameters instead of ~15 foreach kernels each (321 -> ~10
        # launches per step; the step is launch-bound even inside a hipGraph).  Same update rule (train.py:286).
        fused = self.params[0].is_cuda
        self.optimizer = torch.optim.Adam(self.params, lr=opt.lr, weight_decay=opt.weight_decay,
                                          capturable=capturable, fused=fused)
        # clip + step as three launches over a tensor table (gvl_amd/optim.py); torch's own launches wherever that does not apply
        from .optim import ClipAdam
        self.clip_adam = ClipAdam(self.optimizer, opt.grad_clip)
        self.world = world_size

    def _clip_and_step(self):
        active = [p for p in self.params if p.grad is not None]
        if not self.clip_adam.step(active):
            torch.nn.utils.clip_grad_norm_(active, self.opt.grad_clip)
            self.optimizer.step()

    def _forward_loss(self, dt):
        # cache_enabled=False: autocast's weight-cast cache must not live across a hipGraph capture (the cached bf16
        # copies would be graph-pool tensors that the next replay overwrites while autograd still holds them) -- the
        # same precaution torch.cuda.make_graphed_callables takes
        with torch.autocast("cuda", dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None,
                            cache_enabled=False):
            out, loss = self.model(dt, self.criterion, None, self.opt.transformer_input_type)
        from .criterion import weighted_loss_sum
        return weighted_loss_sum(loss, self.criterion.weight_dict, self.__dict__.setdefault("_loss_weights", {})), loss

    def __call__(self, dt):
        self.buckets.zero()                                                    # optimizer.zero_grad(), flat
        final, loss = self._forward_loss(dt)
        with deferred_wgrads():                                                # (the layers' weight gradients in grouped launches)
            final.backward()
        self.buckets.finish()
        hidden = [(p, p.grad) for p in self.buckets.unused_params()]
        for p, _ in hidden:
            p.grad = None                                   # as train.py's loop sees them: Adam skips these
        self._clip_and_step()
        for p, g in hidden:
            p.grad = g
        return final.detach(), _detached(loss)


def _detached(loss):
    """the loss dict without its autograd history: the caller logs these values, and a captured step keeps them for its
    whole life -- attached, they hold the step's graph nodes (the parameters' AccumulateGrad nodes among them) alive, and
    the next capture, on another stream, records waits on those nodes' stream into its graph"""
    return {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in loss.items()}


class _LRU(OrderedDict):
    """captured graphs, least recently used first; each entry pins a private memory pool, so the cache is bounded"""

    def __init__(self, limit):
        super().__init__()
        self.limit = max(1, int(limit))

    def lookup(self, key):
        entry = self.get(key)
        if entry is not None:
            self.move_to_end(key)
        return entry

    def store(self, key, entry):
        self[key] = entry
        self.move_to_end(key)
        while len(self) > self.limit:
            self.popitem(last=False)


_STATIC_KEYS = ("video_tensor", "video_mask", "video_length")


def _drop_superseded(graphs, key, keep_width_buckets=False):
    """grow-only capacities never shrink, so a padded graph of the same tensor shapes with a smaller one is never
    replayed again.  Key layout: ("padded", shapes, slots, cap_len, rows[, epoch]); with cap_len_policy="bucket" the
    caption width is not a grow-only capacity: graphs that differ only in it stay."""
    def grow_part(k):
        return ((k[2],) + tuple(k[4:])) if keep_width_buckets else tuple(k[2:])
    for k in [k for k in graphs if k[0] == "padded" and k[1] == key[1] and grow_part(k) != grow_part(key)]:
        del graphs[k]


class _PaddedBatch:
    """static device buffers of one captured step: the three feature tensors of pdvc.py:250-258 + PaddedTargets"""

    def __init__(self, dt, slots, cap_len, pair_rows=0):
        dev = dt["video_tensor"].device
        self.dt = {k: dt[k].clone() for k in _STATIC_KEYS}
        self.targets = PaddedTargets(dt["video_tensor"].shape[0], slots, cap_len, dev, pair_rows)
        self.dt["_gvl_targets"] = self.targets
        # paths that still want the reference's list (LayerMatch.host(), PostProcess) are served from the host counts
        self.dt["video_target"] = None

    def load(self, dt, num_boxes=None):
        for k in _STATIC_KEYS:
            self.dt[k].copy_(dt[k], non_blocking=True)
        self.targets.load(dt, num_boxes)
        for k, v in dt.items():                               # non-tensor bookkeeping travels along (video_key, cap_raw ...)
            if k not in self.dt and not isinstance(v, torch.Tensor):
                self.dt[k] = v
        self.dt["video_target"] = dt["video_target"]


class _Capacity:
    """Grow-only capacities of the padded layout: slots per video = next power of two >= the largest event count seen
    (>= 4), caption width = next multiple of 4 >= the widest caption tensor seen, caption rows (matched pairs of a
    whole batch, train only) = next multiple of 32 >= the largest event total seen.  A batch that does not fit raises
    the capacity and the step is captured again; in a steady run that happens a handful of times, then never."""

    def __init__(self, slots=0, cap_len=0, pair_rows=0, cap_len_policy="grow", cap_len_step=4):
        self.slots, self.cap_len, self.pair_rows = int(slots or 0), int(cap_len or 0), int(pair_rows or 0)
        self.cap_len_policy, self.cap_len_step = cap_len_policy, max(1, int(cap_len_step))

    def fit(self, dt, with_captions):
        n_gt, cap_len = needed_capacity(dt)
        self.slots = max(self.slots, round_up_pow2(n_gt, 4))
        if not with_captions:
            return self.slots, 0
        step = self.cap_len_step
        bucket = step * ((max(cap_len, 2) + step - 1) // step)
        self.cap_len = max(self.cap_len, bucket)
        self.pair_rows = max(self.pair_rows, 32 * ((max(total_events(dt), 1) + 31) // 32))
        # "bucket": the step runs at THIS batch's caption-width bucket (one graph per bucket of cap_len_step tokens)
        # instead of at the widest caption tensor seen so far -- the teacher-forced loop is serial in
        # the caption length, so a batch of 10-word captions then pays 11 steps, not the 23 a 22-word batch once needed
        return self.slots, (bucket if self.cap_len_policy == "bucket" else self.cap_len)


class GraphedTrainStep(TrainStep):
    """The whole train step -- forward, on-device Hungarian matching, losses, backward, gradient exchange, clipping and
    Adam -- captured in a hipGraph and replayed with a single launch.  The eager step issues ~5 000 kernel launches
    whose host-side cost (~55 ms) exceeds their GPU time (~30 ms); nothing in the step reads the device back
    (gvl_amd.matcher.LayerMatch stays on the device), which is what makes the capture possible.

    **Layout independence.**  The reference loop (train.py:385-409) feeds batches whose number of events per video,
    number of captions and caption lengths differ every iteration.  The captured step therefore runs on
    ``gvl_amd.targets.PaddedTargets``: fixed-shape target / caption buffers with device-side counts, refreshed before
    every replay, so ONE graph serves every batch that fits its capacity (``_Capacity``: grow-only).  The graph cache
    is keyed on tensor shapes and capacities only and is LRU-bounded (``max_graphs``).  Models / criteria outside the
    padded path's domain (``PDVC.supports_padded_targets``, ``'gt_proposals'`` decoder input) fall back to one graph
    per batch layout -- same cache, same bound.

    **Warm-up is side-effect free.**  A capture needs >= 1 eager run first (autograd's accumulators, hipBLASLt
    workspaces, lazily built constants); those runs and the capture itself leave parameters, Adam state, buffers and
    the RNG state exactly as they were (snapshot / restore), so every batch gets ONE update, as in train.py."""

    def __init__(self, model, criterion, opt, world_size=1, process_group=None, warmup=2, split_exchange=None,
                 autocast_dtype=None, max_graphs=16, max_gt=0, max_cap_len=0, max_events=0, padded=None,
                 cap_len_policy="bucket", cap_len_step=2):
        """split_exchange (default: exactly when there is more than one process): the step is captured as THREE graphs --
        (1) zero_grad + forward + losses + backward down to the encoder output, (2) the encoder's backward, (3) clip +
        Adam -- with the bucketed RCCL all-reduce issued eagerly between the replays, so no collective is ever inside a
        hipGraph: the buckets of stage 1 (captioner, heads, decoder) are posted before graph 2 is replayed and travel
        while it runs, the encoder's buckets follow, graph 3 waits for all of them.  The criterion's
        own ``all_reduce(num_boxes)`` (criterion.py:178-180) is taken before the first graph from the host-known
        target counts and reaches the captured criterion through device memory.
        max_gt / max_cap_len / max_events: initial capacities of the padded layout -- events per video, caption tensor
        width, events per batch (0 = grow from the batches seen).
        padded: None = automatic, False = always one graph per batch layout.
        cap_len_policy: "bucket" (default) = one graph per caption-width bucket of `cap_len_step` (2) tokens (<= 15 graphs at
        max_caption_len = 30, hence max_graphs = 16: ~1.5 GB of pool each, spent from 288 GB to save serial token steps), each
        batch replays the graph of its own bucket and pays the teacher-forced steps of ITS longest caption, as the reference
        loop does (LSTM_DSA.py:110-112); "grow" = ONE graph at the widest caption tensor seen (every batch pays the longest
        caption's steps; 5.2 GB of graph pool at cfg A).  Rotating workload: 10.46 ms per step with "grow", 9.67 with buckets
        of 4, 9.50 of 2, 9.46 of 1.  (Rounds 2-3 measured the opposite, 11.8 against 11.0 ms: the captures after
        the first recorded waits on the previous capture's still-alive autograd nodes -- TrainStep returned its loss values
        attached -- and "bucket" replays exactly those later captures.)"""
        from . import graph_replay_guard
        graph_replay_guard("GraphedTrainStep")
        self.split = (world_size > 1) if split_exchange is None else bool(split_exchange)
        # data-parallel form: the backward is cut at the encoder output (`memory`).  Stage 1 (captioner, heads, decoder:
        # ~2/3 of the 100 MB of gradients) completes first; its buckets travel while stage 2 (deformable encoder, base
        # encoder) still runs -- the exchange is eager RCCL between graph replays, never inside a hipGraph
        late = None
        if self.split and hasattr(model, "encoder_decoder_parameters"):
            late = [p for p in model.encoder_decoder_parameters()[1] if p.requires_grad]
        super().__init__(model, criterion, opt, world_size, process_group, capturable=True,
                         flat=True if self.split else None, overlap=not self.split, autocast_dtype=autocast_dtype,
                         first=late)
        self.two_stage = bool(self.split and late and self.buckets.n_first)
        self.time_exchange = self.split                      # three events per step around the eager collectives
        self._exchange_events = []
        self._cut = None
        self.graphs = _LRU(max_graphs)
        self.capacity = _Capacity(round_up_pow2(max_gt, 4) if max_gt else 0, max_cap_len,
                                  32 * ((max_events + 31) // 32) if max_events else 0, cap_len_policy, cap_len_step)
        self.padded = padded
        self.captures = self.replays = 0
        # >= 1 real step must run before the capture (on a side stream): autograd's gradient accumulators, hipBLASLt
        # workspaces and the library's lazily built constants have to exist before a stream is capturing
        self.warmup = max(1, int(warmup))
        # Adam creates its state lazily inside the first step(); if that first step were the captured one, the
        # zero-fills of (step, exp_avg, exp_avg_sq) would become graph nodes and every replay would reset the moments.
        # Create the state now, exactly as torch.optim.Adam._init_group does for capturable=True.
        for p in self.params:
            st = self.optimizer.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    # ---- batch forms -------------------------------------------------------------------------------------------
    @staticmethod
    def _tensor_keys(dt):
        return [k for k, v in dt.items() if isinstance(v, torch.Tensor)]

    def _use_padded(self, dt):
        if self.padded is False or self.opt.transformer_input_type != "queries":
            return False
        slots = max(self.capacity.slots, round_up_pow2(needed_capacity(dt)[0], 4))
        ok = self.model.supports_padded_targets(self.criterion, eval_mode=False, batch=dt["video_tensor"].shape[0],
                                                slots=slots)
        if not ok and self.padded:
            raise RuntimeError("GraphedTrainStep(padded=True): this model / criterion has no padded-target path")
        return ok and slots <= min(64, self.opt.num_queries)

    def _layout_key(self, dt):
        """fallback form: one graph per batch layout (tensor shapes, events per video, teacher-forcing length)"""
        sig = tuple((k, tuple(dt[k].shape), str(dt[k].dtype)) for k in self._tensor_keys(dt))
        n_gt = tuple(len(t_["boxes"]) for t_ in dt["video_target"])
        if "_gvl_cap_steps" not in dt:                      # teacher-forcing length: one host read per NEW batch
            live = (dt["cap_tensor"][:, 1:] != 0).any(0).cpu().tolist()
            dt["_gvl_cap_steps"] = min(1 + (live.index(False) if False in live else len(live)),
                                       dt["cap_tensor"].shape[-1] - 1)
        return "layout", sig, n_gt, dt["_gvl_cap_steps"]

    @staticmethod
    def _static_copy(dt):
        st = dict(dt)
        for k in GraphedTrainStep._tensor_keys(dt):
            st[k] = dt[k].clone()
        st["video_target"] = [{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in t_.items()}
                              for t_ in dt.get("video_target") or []]
        return st

    @staticmethod
    def _refresh(st, dt):
        for k, v in dt.items():
            if isinstance(v, torch.Tensor):
                st[k].copy_(v, non_blocking=True)
        for a, b in zip(st["video_target"], dt.get("video_target") or []):
            for k, v in b.items():
                if isinstance(v, torch.Tensor):
                    a[k].copy_(v, non_blocking=True)

    def _eager(self, dt):
        return TrainStep.__call__(self, dt)

    def _global_num_boxes(self, dt):
        """criterion.py:178-181, outside any capture: mean over the ranks of the (host-known) number of targets, left in
        a persistent DEVICE scalar that the captured criterion reads at replay time -- no host read, and the value is
        not part of the graph key (a float argument re-captured the step for every new value)"""
        n = float(sum(len(t_["labels"]) for t_ in dt["video_target"]))
        nb = self.__dict__.get("_nb_dev")
        if nb is None:
            nb = self._nb_dev = torch.zeros(1, dtype=torch.float32, device=self.params[0].device)
        nb.fill_(n)
        if dist.is_available() and dist.is_initialized() and self.world > 1:
            dist.all_reduce(nb)
            nb.div_(self.world)
        nb.clamp_(min=1.0)
        return nb

    def _forward_backward(self, dt):
        """stage 1 of the data-parallel step: zero_grad, forward, losses, backward down to the encoder output"""
        self.buckets.zero()
        cut = {}

        def memory_cut(memory):
            cut["src"], cut["leaf"] = memory, memory.detach().requires_grad_()
            return cut["leaf"]
        if self.two_stage:
            self.model.memory_cut = memory_cut
        try:
            final, loss = self._forward_loss(dt)
        finally:
            self.model.memory_cut = None
        with deferred_wgrads():
            final.backward()
        self._cut = cut
        return final.detach(), _detached(loss)

    def _backward_encoder(self):
        """stage 2: the gradient of the encoder output through the deformable encoder and the base encoder"""
        if self.two_stage and self._cut:
            with deferred_wgrads():
                self._cut["src"].backward(self._cut["leaf"].grad)
        self._cut = None

    def _update(self, unused=()):
        """clip + Adam over the parameters that take part in this captured step; `unused` (found by the warm-up probe)
        are hidden exactly as TrainStep.__call__ hides them: grad = None while the optimizer looks"""
        hidden = [(p, p.grad) for p in unused]
        for p, _ in hidden:
            p.grad = None
        try:
            self._clip_and_step()
        finally:
            for p, g in hidden:
                p.grad = g

    # ---- side-effect-free warm-up --------------------------------------------------------------------------------
    def _snapshot(self):
        dev = self.params[0].device
        opt_state = [{k: v.clone() for k, v in self.optimizer.state[p].items() if isinstance(v, torch.Tensor)}
                     for p in self.params]
        return ([p.detach().clone() for p in self.params], [b.detach().clone() for b in self.model.buffers()],
                opt_state, torch.cuda.get_rng_state(dev), torch.get_rng_state())

    def _restore(self, snap):
        params, buffers, opt_state, cuda_rng, cpu_rng = snap
        with torch.no_grad():
            for p, v in zip(self.params, params):
                p.copy_(v)                                   # in place: the graphs captured these addresses
            for b, v in zip(self.model.buffers(), buffers):
                b.copy_(v)
            for p, saved in zip(self.params, opt_state):
                for k, v in saved.items():
                    self.optimizer.state[p][k].copy_(v)
        torch.cuda.set_rng_state(cuda_rng, self.params[0].device)
        torch.set_rng_state(cpu_rng)

    def _agree_structural(self, st, dt):
        """layout-keyed data-parallel captures (every rank captures from ITS batch): which parameters may be hidden from the
        captured clip + Adam is agreed ONCE across the ranks from an eager probe step (side-effect free), not per capture --
        GradBuckets.agree_unused.  Until a step where some rank has events, nothing is hidden."""
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            def fb():
                self._forward_backward(st)
                self._backward_encoder()
            unused = self.buckets.probe_unused(fb)
        torch.cuda.current_stream().wait_stream(side)
        self._restore(snap)
        has_events = sum(len(t_["labels"]) for t_ in dt["video_target"]) > 0
        return self.buckets.agree_unused(unused, has_events)

    def _capture(self, st, static_structure=True, hidden=None):
        """warm up on `st` (local work only: no collective, so ranks may capture at different times) and capture;
        -> (graphs, outs).  Parameters / optimizer / RNG are restored afterwards: the capture records, it does not run.
        static_structure: the step's autograd graph does not depend on the batch (the padded form: fixed-capacity target
        and caption rows, masked where empty), so the set of parameters it never reaches is the same on every rank and
        may be hidden from the captured clip + Adam.  A layout-keyed capture is made from THIS rank's batch -- a rank
        whose batch has no events would hide the captioner's parameters while the others apply the averaged gradient --
        so there every parameter stays visible: all ranks update all parameters from the all-reduced buffer and the
        replicas stay identical (the set cannot be agreed inside _capture: captures happen at rank-dependent
        times and must not contain a collective)."""
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        unused = list(hidden) if (hidden and not static_structure) else []     # (agreed across the ranks: _agree_structural)
        with torch.cuda.stream(side):
            for it in range(self.warmup):
                if self.split:
                    def fb():
                        self._forward_backward(st)
                        self._backward_encoder()
                    if it == 0 and self.buckets.flat is not None and static_structure:
                        # which parameters does this step's autograd graph reach?  (static per capture: the padded step
                        # has no data-dependent structure, the layout-keyed step is captured per layout)
                        unused = self.buckets.probe_unused(fb)
                    else:
                        fb()
                    self._update(unused)
                else:
                    TrainStep.__call__(self, st)
        torch.cuda.current_stream().wait_stream(side)
        # capture ON the warm-up stream: autograd's AccumulateGrad nodes were created there and keep running there;
        # captured from another stream they become a parallel branch of the graph that the optimizer kernels do not
        # wait for (seen as NaN parameters after a few replays of the bf16 step)
        if self.split:
            g_fb, g_enc, g_up = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_fb, stream=side):
                outs = self._forward_backward(st)
            with torch.cuda.graph(g_enc, pool=g_fb.pool(), stream=side):
                self._backward_encoder()
            with torch.cuda.graph(g_up, pool=g_fb.pool(), stream=side):
                self._update(unused)
            self.last_unused = unused
            graphs = (g_fb, g_enc, g_up)
        else:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                outs = TrainStep.__call__(self, st)
            graphs = (graph,)
        self._restore(snap)
        self.captures += 1
        return graphs, outs

    def _mean_num_boxes_on_device(self, pt):
        """criterion.py:178-181 without a host round trip: load() left this rank's target count in pt.num_boxes; sum over
        the ranks, divide, clamp -- three tiny stream-ordered operations on the device scalar the captured criterion reads"""
        if dist.is_available() and dist.is_initialized() and self.world > 1:
            pt.num_boxes.copy_(pt.counts.sum())                    # un-clamped local count
            dist.all_reduce(pt.num_boxes)
            pt.num_boxes.div_(self.world).clamp_(min=1.0)

    def __call__(self, dt):
        padded = self._use_padded(dt)
        # fallback form only: the cross-rank mean of the target count, in a device scalar (one small collective per step)
        nb = self._global_num_boxes(dt) if (self.split and not padded) else None
        if padded:
            slots, cap_len = self.capacity.fit(dt, with_captions=True)
            rows = min(self.capacity.pair_rows, dt["video_tensor"].shape[0] * min(slots, self.opt.num_queries))
            key = ("padded", tuple((k, tuple(dt[k].shape), str(dt[k].dtype)) for k in _STATIC_KEYS), slots, cap_len, rows)
            entry = self.graphs.lookup(key)
            if entry is None:
                _drop_superseded(self.graphs, key, self.capacity.cap_len_policy == "bucket")
                batch = _PaddedBatch(dt, slots, cap_len, rows)
                batch.load(dt)
                graphs, outs = self._capture(batch.dt)
                entry = (graphs, batch, outs)
                self.graphs.store(key, entry)
            graphs, batch, outs = entry
            batch.load(dt)
            if self.split:
                self._mean_num_boxes_on_device(batch.targets)       # every rank, every step: one scalar all-reduce
        else:
            key = self._layout_key(dt)
            entry = self.graphs.lookup(key)
            self.criterion.num_boxes_override = nb              # device scalar read by the captured criterion
            try:
                if (self.split and self.world > 1 and self.buckets.flat is not None
                        and self.__dict__.get("_structural") is None):
                    # a collective, so NOT tied to this rank's cache miss: every rank runs it on every layout-keyed step until
                    # the set is agreed (the outcome -- agreed / no rank had events yet -- is the same on all of them)
                    self._structural = self._agree_structural(GraphedTrainStep._static_copy(dt), dt)
                    if self._structural is not None:
                        self.graphs.clear()                      # (captured while nothing was hidden: capture again)
                        entry = None
                if entry is None:
                    st = GraphedTrainStep._static_copy(dt)
                    graphs, outs = self._capture(st, static_structure=False, hidden=self.__dict__.get("_structural"))
                    entry = (graphs, st, outs)
                    self.graphs.store(key, entry)
            finally:
                self.criterion.num_boxes_override = None
            graphs, st, outs = entry
            self._refresh(st, dt)
        graphs[0].replay()
        if self.split:
            # eager RCCL between the replays: the first-stage buckets are on the wire while the encoder's backward runs
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if self.time_exchange else None
            if ev:
                ev[0].record()
            self.buckets.exchange_begin(0)
            graphs[1].replay()
            if ev:
                ev[1].record()                                   # encoder backward queued: what follows is exposed
            self.buckets.exchange_begin(1)
            self.buckets.exchange_end()
            if ev:
                ev[2].record()
                self._exchange_events.append(ev)
                if len(self._exchange_events) > 64:
                    self._exchange_events.pop(0)
            graphs[2].replay()
        # a replay updates the parameters through the addresses the graph recorded: torch's version counters do not see it, and
        # every weight-derived cache (operand planes, captured eval / decode graphs) is keyed on them
        bump_versions(self.params)
        self.replays += 1
        return outs

    def exchange_times_ms(self):
        """(total, exposed) milliseconds of the gradient exchange per step, means over the recorded steps: total = first
        bucket posted -> all buckets reduced and averaged (it overlaps the encoder's backward), exposed = the part after
        the encoder's backward graph was queued.  Synchronises; call it outside the timed region."""
        if not self._exchange_events:
            return None
        torch.cuda.synchronize()
        tot = [e[0].elapsed_time(e[2]) for e in self._exchange_events]
        exp = [e[1].elapsed_time(e[2]) for e in self._exchange_events]
        self._exchange_events = []
        return sum(tot) / len(tot), sum(exp) / len(exp)


class GraphedEvalForward:
    """The whole evaluation forward -- base encoder, deformable encoder / decoder, heads, greedy captioning loop, on-device
    matching and the set criterion -- captured in hipGraphs and replayed.  The eager forward is host-bound outside the
    captioner (≈450 launches, 4.9 ms for 1.5 ms of GPU work at cfg A).

    **Layout independence**: as in GraphedTrainStep, the ground truth travels as ``PaddedTargets`` (device-side counts),
    so batches with different numbers of events per video (eval_utils.py:187-203 feeds them in dataset order) replay the
    same graph; the cache is keyed on tensor shapes + capacity + parameter version and LRU-bounded.

    **Finished captions stop costing**: the reference leaves the decoding loop at the first step where every sequence
    has ended (LSTM_DSA.py:186-187).  The loop is captured in segments of ``decode_chunk`` tokens; after each segment
    ONE small device->host read of the `alive` flags decides whether the next segment is replayed.  Trained
    checkpoints (captions of ~10-15 tokens against max_caption_len = 30) therefore pay for the longest caption of the
    batch, rounded up to the chunk, not for 30 steps.  decode_chunk=0: one graph, flags read once at the end.
    Returned tensors are the graphs' static outputs: they are overwritten by the next call (clone what must survive)."""

    def __init__(self, model, criterion, transformer_input_type="queries", warmup=1, autocast_dtype=None,
                 max_graphs=4, max_gt=0, decode_chunk=5, padded=None):
        from . import graph_replay_guard
        graph_replay_guard("GraphedEvalForward")
        self.model, self.criterion, self.kind = model, criterion, transformer_input_type
        self.autocast_dtype = autocast_dtype          # e.g. torch.bfloat16: capture the forward under torch.autocast
        self.warmup = max(1, int(warmup))
        self.graphs = _LRU(max_graphs)
        self.capacity = _Capacity(round_up_pow2(max_gt, 4) if max_gt else 0, 0)
        self.decode_chunk = max(0, int(decode_chunk))
        self.padded = padded
        self.captures = self.replays = 0
        self.segments_replayed = 0                    # decode segments actually run (diagnostic)

    def _epoch(self):
        # operands derived from weights only (concatenated / pre-multiplied matrices of the captioner) are cached per
        # parameter version and are constants of the captured graph: updated parameters => a new capture
        ps = self.__dict__.get("_param_list")
        if ps is None:                                        # (module traversal once, not on every call)
            ps = self._param_list = list(self.model.parameters())
        return sum(p_._version for p_ in ps)

    def _use_padded(self, dt):
        if self.padded is False or self.kind != "queries" or self.criterion is None:
            return False
        slots = max(self.capacity.slots, round_up_pow2(needed_capacity(dt)[0], 4))
        return (self.model.supports_padded_targets(self.criterion, eval_mode=True, batch=dt["video_tensor"].shape[0],
                                                   slots=slots)
                and slots <= min(64, self.model.opt.num_queries))

    def _autocast(self):
        # inference under autocast follows GVL_AUTOCAST_INFERENCE (gvl_amd/pdvc.py): the fp32-storage path with one ("f16",
        # default) or three ("fp32") fp16 products per fp32 product, or the bf16-storage path ("bf16"); the decode segments
        # captured after the main forward follow the same policy as PDVC.forward itself
        import contextlib
        from . import MultiScaleDeformableAttention as MSDA
        from .pdvc import autocast_inference_policy, autocast_products
        on = self.autocast_dtype is not None and autocast_inference_policy() == "bf16"
        ctx = contextlib.ExitStack()
        ctx.enter_context(torch.autocast("cuda", dtype=self.autocast_dtype or torch.bfloat16, enabled=on))
        if self.autocast_dtype is not None and not on:
            ctx.enter_context(MSDA.f16_products(autocast_products()))
        return ctx

    def _forward(self, dt):
        with self._autocast():
            return self.model(dt, self.criterion, None, self.kind, eval_mode=True)

    def _continue(self, head, a, b):
        with self._autocast():
            return head.decode_continue(a, b)

    @staticmethod
    def _matches(out):
        found = []
        for o in [out] + list(out.get("aux_outputs", [])):
            m = o.get("matched_indices")
            if hasattr(m, "plan"):
                found.append(m)
        return found

    def _segments(self, head):
        """iteration ranges of the greedy loop: [0, b1) inside the main graph, then one graph per further range"""
        T = head.max_caption_len
        if (self.decode_chunk <= 0 or self.decode_chunk >= T or self.model.opt.eval_disable_captioning
                or self.model.opt.caption_decoder_type != "standard"):
            return [T + 1]
        bounds = list(range(self.decode_chunk + 1, T + 1, self.decode_chunk))
        return bounds + [T + 1]

    def _capture(self, st):
        heads = list(self.model.caption_head)
        head = heads[-1]
        bounds = self._segments(head)
        for h_ in heads:
            h_.defer_trim = True
            h_.decode_stop = bounds[0]
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(self.warmup):
                    self._forward(st)
                    for a, b in zip(bounds[:-1], bounds[1:]):
                        self._continue(head, a, b)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                out, loss = self._forward(st)
                alive0 = getattr(head, "last_alive", None)
            flags = [alive0[:bounds[0] - 1] if alive0 is not None and len(bounds) > 1 else alive0]
            seg_graphs = []
            for a, b in zip(bounds[:-1], bounds[1:]):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=graph.pool(), stream=side):
                    flags.append(self._continue(head, a, b))
                seg_graphs.append(g)
        finally:
            for h_ in heads:
                h_.defer_trim = False
                h_.decode_stop = None
                h_._decode_state = None
        self.captures += 1
        return graph, seg_graphs, out, loss, flags

    @torch.no_grad()
    def __call__(self, dt):
        epoch = self._epoch()
        if self.graphs and next(iter(self.graphs))[-1] != epoch:
            self.graphs.clear()                              # parameters changed: graphs of the old weights are dead
        padded = self._use_padded(dt)
        if padded:
            slots, _ = self.capacity.fit(dt, with_captions=False)
            key = ("padded", tuple((k, tuple(dt[k].shape), str(dt[k].dtype)) for k in _STATIC_KEYS), slots, epoch)
        else:
            sig = tuple((k, tuple(v.shape), str(v.dtype)) for k, v in dt.items() if isinstance(v, torch.Tensor))
            key = ("layout", sig, tuple(len(t_["boxes"]) for t_ in dt.get("video_target") or []), epoch)
        entry = self.graphs.lookup(key)
        if entry is None:
            if padded:
                _drop_superseded(self.graphs, key)
                batch = _PaddedBatch(dt, slots, 0)
                batch.load(dt)
                st = batch.dt
            else:
                batch, st = None, GraphedTrainStep._static_copy(dt)
            entry = (batch, st) + self._capture(st)
            self.graphs.store(key, entry)
        batch, st, graph, seg_graphs, out, loss, flags = entry
        if batch is not None:
            batch.load(dt)
        else:
            GraphedTrainStep._refresh(st, dt)
        graph.replay()
        self.replays += 1
        out = dict(out)
        for m in self._matches(out):
            m.invalidate()                                   # the device indices changed under the cached host copy
        if flags[0] is not None and isinstance(out.get("seq"), torch.Tensor) and not self.model.opt.eval_disable_captioning:
            # The forward's only host reads (LSTM_DSA.py:186-187): the `alive` flags of every decode segment that ran.
            # They are read ONE SEGMENT BEHIND the GPU: segment k + 1 is already queued when the host waits for the flags
            # of segment k (a stream-ordered copy into pinned memory + an event), so the device never idles on the read;
            # the price is at most one speculative segment after the last caption has ended -- it only writes columns
            # that are trimmed below.
            def post(f_):
                host = torch.empty(f_.shape, dtype=f_.dtype, pin_memory=True)
                host.copy_(f_, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                return host, ev

            def take(p_):
                p_[1].synchronize()
                return p_[0].tolist()
            pending = post(flags[0])
            alive, nxt = [], 0
            while True:
                ahead = None
                if nxt < len(seg_graphs):                       # keep the GPU one segment ahead of the flags in hand
                    seg_graphs[nxt].replay()
                    self.segments_replayed += 1
                    ahead = post(flags[nxt + 1])
                    nxt += 1
                alive += take(pending)
                if ahead is None:
                    break
                if False in alive:                              # every caption ended: the segment in flight is the last one
                    take(ahead)
                    break
                pending = ahead
            keep = alive.index(False) if False in alive else len(alive)
            if keep == 0:
                out["seq"], out["caption_probs"] = [], {"cap_prob_eval": []}
            else:
                out["seq"] = out["seq"][..., :keep]
                out["caption_probs"] = {"cap_prob_eval": out["caption_probs"]["cap_prob_eval"][..., :keep]}
        return out, loss
