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

    def __init__(self, params, bucket_bytes=25 << 20, process_group=None, flat=None, overlap=True):
        """flat=None: flat buffer exactly when there is more than one process.  overlap=False: no autograd hooks --
        the buckets are exchanged by an explicit ``exchange()`` after backward (what a captured forward/backward
        needs: collectives stay outside the hipGraph)."""
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # One process, nothing to exchange: leave .grad = None so that autograd hands every gradient over without the
        # per-parameter accumulate kernel into a pre-existing buffer (133 launches per step; the step is launch-bound).
        self.flat = None
        self.buckets, self.bucket_of, self.ready, self.handles = [], {}, [], []
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
        if self.world > 1 and overlap:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)

    def _hook(self, p):
        b = self.bucket_of[id(p)]
        self.ready[b] += 1
        if self.ready[b] == self.buckets[b][2]:
            s, e, _ = self.buckets[b]
            self.handles.append(dist.all_reduce(self.flat[s:e], group=self.group, async_op=True))

    def zero(self):
        if self.flat is None:
            for p in self.params:
                p.grad = None
            return
        self.flat.zero_()
        self.ready = [0] * len(self.buckets)
        self.handles = []

    def exchange(self):
        """non-overlapped variant: all-reduce every bucket now (on the current stream's ordering) and average"""
        if self.world == 1 or self.flat is None:
            return
        handles = [dist.all_reduce(self.flat[s:e], group=self.group, async_op=True) for s, e, _ in self.buckets]
        for h in handles:
            h.wait()
        self.flat.div_(self.world)

    def finish(self):
        """wait for the in-flight buckets, reduce the ones whose parameters got no gradient this step, average."""
        if self.world == 1 or self.flat is None:
            return
        if not self.overlap:
            return self.exchange()
        for b, (s, e, n) in enumerate(self.buckets):
            if self.ready[b] != n:                       # some parameter unused this step: reduce the bucket now
                self.handles.append(dist.all_reduce(self.flat[s:e], group=self.group, async_op=True))
        for h in self.handles:
            h.wait()
        self.flat.div_(self.world)


class TrainStep:
    """zero_grad -> forward -> weighted loss -> backward (+ overlapped all-reduce) -> clip -> Adam (train.py:385-409)"""

    def __init__(self, model, criterion, opt, world_size=1, process_group=None, capturable=False, flat=None,
                 overlap=True, autocast_dtype=None):
        """autocast_dtype (e.g. torch.bfloat16): forward + losses under torch.autocast -- bf16 GEMMs and bf16-storage
        deformable attention, fp32 master weights / Adam, fp32 islands for the teacher-forced captioner loop and the
        criterion kernels."""
        self.model, self.criterion, self.opt = model, criterion, opt
        self.autocast_dtype = autocast_dtype
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.buckets = GradBuckets(self.params, process_group=process_group, flat=flat, overlap=overlap)
        # fused=True: one multi-tensor kernel per ~100 parameters instead of ~15 foreach kernels each (321 -> ~10
        # launches per step; the step is launch-bound even inside a hipGraph).  Same update rule (train.py:286).
        fused = self.params[0].is_cuda
        self.optimizer = torch.optim.Adam(self.params, lr=opt.lr, weight_decay=opt.weight_decay,
                                          capturable=capturable, fused=fused)
        self.world = world_size

    def _forward_loss(self, dt):
        # cache_enabled=False: autocast's weight-cast cache must not live across a hipGraph capture (the cached bf16
        # copies would be graph-pool tensors that the next replay overwrites while autograd still holds them) -- the
        # same precaution torch.cuda.make_graphed_callables takes
        with torch.autocast("cuda", dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None,
                            cache_enabled=False):
            out, loss = self.model(dt, self.criterion, None, self.opt.transformer_input_type)
        wd = self.criterion.weight_dict
        return sum(loss[k].float() * wd[k] for k in loss.keys() if k in wd), loss

    def __call__(self, dt):
        self.buckets.zero()                                                    # optimizer.zero_grad(), flat
        final, loss = self._forward_loss(dt)
        final.backward()
        self.buckets.finish()
        torch.nn.utils.clip_grad_norm_(self.params, self.opt.grad_clip)
        self.optimizer.step()
        return final.detach(), loss


class GraphedTrainStep(TrainStep):
    """The whole train step -- forward, on-device Hungarian matching, losses, backward, gradient exchange, clipping and
    Adam -- captured ONCE per batch layout in a hipGraph and replayed with a single launch.  The eager step issues
    ~5 000 kernel launches whose host-side cost (~55 ms) exceeds their GPU time (~30 ms); nothing in the step reads
    the device back (gvl_amd.matcher.LayerMatch stays on the device), which is what makes the capture possible.

    Inputs are copied into static buffers before every replay; the batch layout (tensor shapes, number of events per
    video, teacher-forcing length) is the cache key."""

    def __init__(self, model, criterion, opt, world_size=1, process_group=None, warmup=3, split_exchange=None,
                 autocast_dtype=None):
        """split_exchange (default: exactly when there is more than one process): the step is captured as TWO graphs --
        zero_grad + forward + backward into the flat gradient buffer, then clip + Adam -- with the bucketed RCCL
        all-reduce issued eagerly between the two replays, so no collective is ever inside a hipGraph.  The criterion's
        own ``all_reduce(num_boxes)`` (criterion.py:178-180) is taken before the first graph from the host-known
        target counts."""
        self.split = (world_size > 1) if split_exchange is None else bool(split_exchange)
        super().__init__(model, criterion, opt, world_size, process_group, capturable=True,
                         flat=True if self.split else None, overlap=not self.split, autocast_dtype=autocast_dtype)
        self.graphs = {}
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

    @staticmethod
    def _tensor_keys(dt):
        return [k for k, v in dt.items() if isinstance(v, torch.Tensor)]

    def _key(self, dt):
        sig = tuple((k, tuple(dt[k].shape), str(dt[k].dtype)) for k in self._tensor_keys(dt))
        n_gt = tuple(len(t_["boxes"]) for t_ in dt["video_target"])
        if "_gvl_cap_steps" not in dt:                      # teacher-forcing length: one host read per NEW batch
            live = (dt["cap_tensor"][:, 1:] != 0).any(0).cpu().tolist()
            dt["_gvl_cap_steps"] = min(1 + (live.index(False) if False in live else len(live)),
                                       dt["cap_tensor"].shape[-1] - 1)
        return sig, n_gt, dt["_gvl_cap_steps"]

    @staticmethod
    def _static_copy(dt):
        st = dict(dt)
        for k in GraphedTrainStep._tensor_keys(dt):
            st[k] = dt[k].clone()
        st["video_target"] = [{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in t_.items()}
                              for t_ in dt["video_target"]]
        return st

    @staticmethod
    def _refresh(st, dt):
        for k, v in dt.items():
            if isinstance(v, torch.Tensor):
                st[k].copy_(v, non_blocking=True)
        for a, b in zip(st["video_target"], dt["video_target"]):
            for k, v in b.items():
                if isinstance(v, torch.Tensor):
                    a[k].copy_(v, non_blocking=True)

    def _eager(self, dt):
        return TrainStep.__call__(self, dt)

    def _global_num_boxes(self, dt):
        """criterion.py:178-181, outside any capture: mean over the ranks of the (host-known) number of targets"""
        n = float(sum(len(t_["labels"]) for t_ in dt["video_target"]))
        if dist.is_available() and dist.is_initialized() and self.world > 1:
            nb = torch.tensor([n], dtype=torch.float32, device=self.params[0].device)
            dist.all_reduce(nb)
            n = float(nb.item()) / self.world
        return max(n, 1.0)

    def _forward_backward(self, dt):
        self.buckets.zero()
        final, loss = self._forward_loss(dt)
        final.backward()
        return final.detach(), loss

    def _update(self):
        torch.nn.utils.clip_grad_norm_(self.params, self.opt.grad_clip)
        self.optimizer.step()

    def _call_split(self, dt):
        nb = self._global_num_boxes(dt)
        key = self._key(dt) + (nb,)
        entry = self.graphs.get(key)
        self.criterion.num_boxes_override = nb                # a kernel argument of the captured criterion
        try:
            if entry is None:
                st = GraphedTrainStep._static_copy(dt)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(self.warmup):
                        self._forward_backward(st)
                        self.buckets.exchange()
                        self._update()
                torch.cuda.current_stream().wait_stream(side)
                g_fb, g_up = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_fb, stream=side):
                    outs = self._forward_backward(st)
                with torch.cuda.graph(g_up, pool=g_fb.pool(), stream=side):
                    self._update()
                entry = self.graphs[key] = (g_fb, g_up, st, outs)
            g_fb, g_up, st, outs = entry
            self._refresh(st, dt)
            g_fb.replay()
            self.buckets.exchange()                           # eager RCCL between the two replays
            g_up.replay()
        finally:
            self.criterion.num_boxes_override = None
        return outs

    def __call__(self, dt):
        if self.split:
            return self._call_split(dt)
        key = self._key(dt)
        entry = self.graphs.get(key)
        if entry is None:
            st = GraphedTrainStep._static_copy(dt)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(self.warmup):               # real optimisation steps (Adam state, lazy inits, LDS attrs)
                    self._eager(st)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # capture ON the warm-up stream: autograd's AccumulateGrad nodes were created there and keep running there;
            # captured from another stream they become a parallel branch of the graph that the optimizer kernels do not
            # wait for (seen as NaN parameters after a few replays of the bf16 step)
            with torch.cuda.graph(graph, stream=side):
                outs = self._eager(st)
            entry = self.graphs[key] = (graph, st, outs)        # capture records, it does not execute: replay below
        graph, st, outs = entry
        self._refresh(st, dt)
        graph.replay()
        return outs


class GraphedEvalForward:
    """The whole evaluation forward -- base encoder, deformable encoder / decoder, heads, greedy captioning loop, on-device
    matching and the set criterion -- captured once per batch layout in a hipGraph and replayed with one launch.
    The eager forward is host-bound outside the captioner (≈450 launches, 4.9 ms for 1.5 ms of GPU work at cfg A).

    The one data-dependent host decision of the reference's eval forward -- cutting the caption tensor at the first
    step where every sequence has ended (LSTM_DSA.py:186-187) -- is taken AFTER the replay from the `alive` flags the
    graph leaves on the device.  Returned tensors are the graph's static outputs: they are overwritten by the next
    call (clone what must survive)."""

    def __init__(self, model, criterion, transformer_input_type="queries", warmup=1, autocast_dtype=None):
        self.model, self.criterion, self.kind = model, criterion, transformer_input_type
        self.autocast_dtype = autocast_dtype          # e.g. torch.bfloat16: capture the forward under torch.autocast
        self.warmup = max(1, int(warmup))
        self.graphs = {}

    def _key(self, dt):
        sig = tuple((k, tuple(v.shape), str(v.dtype)) for k, v in dt.items() if isinstance(v, torch.Tensor))
        # operands derived from weights only (concatenated / pre-multiplied matrices of the captioner) are cached per
        # parameter version and are constants of the captured graph: updated parameters => a new capture
        epoch = sum(p_._version for p_ in self.model.parameters())
        return sig, tuple(len(t_["boxes"]) for t_ in dt["video_target"]), epoch

    def _forward(self, dt):
        with torch.autocast("cuda", dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None):
            return self.model(dt, self.criterion, None, self.kind, eval_mode=True)

    @staticmethod
    def _matches(out):
        found = []
        for o in [out] + list(out.get("aux_outputs", [])):
            m = o.get("matched_indices")
            if hasattr(m, "plan"):
                found.append(m)
        return found

    @torch.no_grad()
    def __call__(self, dt):
        key = self._key(dt)
        if self.graphs and next(iter(self.graphs))[-1] != key[-1]:
            self.graphs.clear()                              # parameters changed: graphs of the old weights are dead
        entry = self.graphs.get(key)
        heads = list(self.model.caption_head)
        if entry is None:
            st = GraphedTrainStep._static_copy(dt)
            for h_ in heads:
                h_.defer_trim = True
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(self.warmup):
                        self._forward(st)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    out, loss = self._forward(st)
                alive = getattr(heads[-1], "last_alive", None)
            finally:
                for h_ in heads:
                    h_.defer_trim = False
            entry = self.graphs[key] = (graph, st, out, loss, alive)
        graph, st, out, loss, alive = entry
        GraphedTrainStep._refresh(st, dt)
        graph.replay()
        out = dict(out)
        for m in self._matches(out):
            m._host = None                                   # the device indices changed under the cached host copy
        if alive is not None and isinstance(out.get("seq"), torch.Tensor) and not self.model.opt.eval_disable_captioning:
            flags = alive.cpu().tolist()                     # the forward's only host read (LSTM_DSA.py:186-187)
            keep = flags.index(False) if False in flags else len(flags)
            if keep == 0:
                out["seq"], out["caption_probs"] = [], {"cap_prob_eval": []}
            else:
                out["seq"] = out["seq"][..., :keep]
                out["caption_probs"] = {"cap_prob_eval": out["caption_probs"]["cap_prob_eval"][..., :keep]}
        return out, loss
