"""CPU, world_size = 2, gloo: the data-parallel machinery of gvl_amd.parallel (flat-buffer gradient buckets with
overlapped all-reduce, video sharding) and the criterion's all_reduce(num_boxes) (criterion.py:178-180).  The HIP
kernels cannot run here, so the model in these tests is a small pure-PyTorch module; the machinery is model-agnostic."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
                               torch.nn.Linear(32, 3))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gvl_amd.parallel import GradBuckets
        from gvl_amd.criterion import SetCriterion
        model = _make_model()
        buckets = GradBuckets(list(model.parameters()), bucket_bytes=1024)       # several buckets
        assert len(buckets.buckets) >= 2
        g = torch.Generator().manual_seed(5)
        x = torch.randn(8, 12, generator=g)
        y = torch.randn(8, 3, generator=g)
        xs, ys = x[rank::world], y[rank::world]
        for _ in range(2):                                                       # second step: buffers are reused
            buckets.zero()
            loss = ((model(xs) - ys) ** 2).sum() / 8 * world                     # so that the rank-mean is the global loss
            loss.backward()
            buckets.finish()
        grads = [p.grad.clone() for p in model.parameters()]
        assert all(p.grad.data_ptr() >= buckets.flat.data_ptr() for p in model.parameters())
        # the non-overlapped form used between two graph replays (GraphedTrainStep, split_exchange): same gradients
        model2 = _make_model()
        b2 = GradBuckets(list(model2.parameters()), bucket_bytes=1024, flat=True, overlap=False)
        b2.zero()
        (((model2(xs) - ys) ** 2).sum() / 8 * world).backward()
        b2.exchange()
        for a_, p_ in zip(grads, model2.parameters()):
            assert torch.allclose(a_, p_.grad, atol=1e-7)
        # criterion with the cross-rank mean of num_boxes supplied by the caller (no collective inside)

        # criterion: num_boxes is summed over ranks then divided by world (criterion.py:178-181)
        class M(torch.nn.Module):
            def forward(self, outputs, targets):
                n = [len(t_["boxes"]) for t_ in targets]
                idx = [(torch.arange(k), torch.arange(k)) for k in n]
                return idx, idx
        import argparse
        crit = SetCriterion(1, M(), {}, ['boxes'], opt=argparse.Namespace(lloss_gau_mask=1, lloss_beta=1))
        nb = 3 if rank == 0 else 1
        tg = [{"boxes": torch.tensor([[0.5, 0.2]] * nb), "labels": torch.zeros(nb, dtype=torch.long)}]
        outs = {"pred_logits": torch.zeros(1, 4, 1), "pred_boxes": torch.tensor([[[0.4, 0.2]] * 4]),
                "pred_count": torch.zeros(1, 11)}
        losses, _ = crit(outs, tg)
        crit.num_boxes_override = 2.0
        losses_o, _ = crit(outs, tg)
        assert abs(float(losses_o["loss_bbox"]) - float(losses["loss_bbox"])) < 1e-7
        q.put((rank, [g_.numpy() for g_ in grads], float(losses["loss_bbox"])))
    except Exception:                                    # surface the failure instead of letting the parent time out
        import traceback
        q.put((rank, "error", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_gradient_buckets_allreduce_equals_single_process():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for r in res:
        assert r[1] != "error", r[2]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    # single-process reference on the full batch
    model = _make_model()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 12, generator=g)
    y = torch.randn(8, 3, generator=g)
    (((model(x) - y) ** 2).sum() / 8).backward()
    for i, p in enumerate(model.parameters()):
        for r in range(world):
            assert torch.allclose(torch.from_numpy(res[r][1][i]), p.grad, atol=1e-6), (r, i)
    # loss_bbox = sum|src - tgt| / num_boxes, num_boxes = (3 + 1) / 2 = 2 on both ranks
    assert abs(res[0][2] - 3 * 0.1 / 2) < 1e-6 and abs(res[1][2] - 1 * 0.1 / 2) < 1e-6


def test_shard_batch_partitions_videos_and_captions():
    from gvl_amd.parallel import shard_batch
    B = 5
    n_gt = [2, 1, 3, 1, 2]
    dt = {"video_tensor": torch.arange(B)[:, None, None].float().expand(B, 4, 2),
          "video_mask": torch.ones(B, 4, dtype=torch.bool), "video_length": torch.arange(B)[:, None].float().expand(B, 3),
          "gt_boxes_mask": torch.ones(B, 3, dtype=torch.bool),
          "video_target": [{"boxes": torch.zeros(n, 2), "labels": torch.zeros(n)} for n in n_gt],
          "cap_raw": [["c"] * n for n in n_gt],
          "cap_tensor": torch.arange(sum(n_gt))[:, None].expand(-1, 6), "cap_mask": torch.ones(sum(n_gt), 6)}
    seen_v, seen_c = [], []
    for r in range(2):
        s = shard_batch(dt, r, 2)
        seen_v += s["video_tensor"][:, 0, 0].tolist()
        seen_c += s["cap_tensor"][:, 0].tolist()
        assert len(s["video_target"]) == s["video_tensor"].shape[0] == len(s["cap_raw"])
        assert s["cap_tensor"].shape[0] == sum(len(t_["boxes"]) for t_ in s["video_target"])
    assert sorted(seen_v) == list(range(B)) and sorted(seen_c) == list(range(sum(n_gt)))
    # B not a multiple of W: the leading ranks hold one video more; a rank without a video gets None
    sizes = [shard_batch(dt, r, 3)["video_tensor"].shape[0] for r in range(3)]
    assert sum(sizes) == B and max(sizes) - min(sizes) <= 1
    assert shard_batch(dt, B, B + 1) is None and shard_batch(dt, B - 1, B + 1)["video_tensor"].shape[0] == 1


def _worker_gather(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gvl_amd.eval_utils import gather_results, shard_indices
        from gvl_amd.parallel import GradBuckets
        keys = [f"v_{i:03d}" for i in range(7)]
        mine = {keys[i]: [{"timestamp": [float(i), float(i) + 1.5], "sentence": f"s{i}", "query_id": i % 3}
                          for _ in range(1 + i % 2)] for i in shard_indices(len(keys), rank, world)}
        merged = gather_results(mine, dst=0)
        # ---- bucket launch order: rank 1 completes its buckets in the opposite order and leaves one parameter
        # without a gradient; both ranks must still post the same sequence of collectives (parallel.py:80)
        torch.manual_seed(0)
        layers = [torch.nn.Linear(16, 16) for _ in range(4)]
        params = [p for l_ in layers for p in l_.parameters()]
        buckets = GradBuckets(params, bucket_bytes=600, overlap=True)
        assert len(buckets.buckets) >= 3
        x = torch.ones(2, 16) * (rank + 1)
        order = layers if rank == 0 else list(reversed(layers))
        buckets.zero()
        used = order[:3]                                     # the 4th layer of each rank's order gets no gradient locally
        loss = sum(l_(x).sum() for l_ in used)
        loss.backward()
        buckets.finish()
        grads = [p.grad.clone().numpy() for p in params]
        unused = [i for i, p in enumerate(params) if any(p is u for u in buckets.unused_params())]
        # a second step in which BOTH ranks leave layer 3 (and each rank one more, different, layer) without a gradient:
        # only the layer nobody used is hidden from the optimizer, on every rank (the set is agreed)
        buckets.zero()
        sum(l_(x).sum() for l_ in (layers[:2] if rank == 0 else layers[1:3])).backward()
        buckets.finish()
        unused += [100 + i for i, p in enumerate(params) if any(p is u for u in buckets.unused_params())]
        # the captured (no-hook) form finds the parameters a step never reaches with a probe run
        probe = GradBuckets([p for l_ in layers for p in l_.parameters()], flat=True, overlap=False)
        never = probe.probe_unused(lambda: sum(l_(x).sum() for l_ in layers[:3]).backward())
        unused += [200 + i for i, p in enumerate(probe.params) if any(p is u for u in never)]
        # ---- two-stage exchange of the captured data-parallel step: the `first` group (gradients complete after the
        # first backward stage) occupies the leading buckets and travels while the second stage would still run
        torch.manual_seed(1)
        l2 = [torch.nn.Linear(16, 16) for _ in range(4)]
        p2 = [p for l_ in l2 for p in l_.parameters()]
        late = [p for l_ in l2[2:] for p in l_.parameters()]
        b2 = GradBuckets(p2, bucket_bytes=600, flat=True, overlap=False, first=late)
        assert 0 < b2.n_first < len(b2.buckets)
        lead = {b2.bucket_of[id(p)] for p in late}
        assert lead == set(range(b2.n_first)) and all(b2.bucket_of[id(p)] >= b2.n_first for p in p2 if all(p is not q_ for q_ in late))
        b2.zero()
        xx = torch.ones(2, 16) * (rank + 1)
        sum(l_(xx).sum() for l_ in l2[2:]).backward()                  # stage 1: the late layers
        b2.exchange_begin(0)
        sum(l_(xx).sum() for l_ in l2[:2]).backward()                  # stage 2 overlaps the first exchange
        b2.exchange_begin(1)
        b2.exchange_end()
        two_stage = [p.grad.clone().numpy() for p in p2]
        q.put((rank, merged, grads, unused, two_stage))
    except Exception:
        import traceback
        q.put((rank, "error", traceback.format_exc(), None, None))
    finally:
        dist.destroy_process_group()


def test_eval_result_gather_and_fixed_bucket_order():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_gather, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for r in res:
        assert r[1] != "error", r[2]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    merged0, merged1 = res[0][1], res[1][1]
    assert merged1 is None and sorted(merged0) == [f"v_{i:03d}" for i in range(7)]
    assert [len(merged0[f"v_{i:03d}"]) for i in range(7)] == [1 + i % 2 for i in range(7)]
    assert merged0["v_003"][0]["sentence"] == "s3"
    # gradients: mean over the ranks of each rank's local gradient (zeros where a rank had none)
    torch.manual_seed(0)
    layers = [torch.nn.Linear(16, 16) for _ in range(4)]
    want = []
    for li, l_ in enumerate(layers):
        gw = sum((r + 1.0) * (1.0 if (li < 3 if r == 0 else li > 0) else 0.0) for r in range(2)) * 2 / 2
        gb = sum((1.0 if (li < 3 if r == 0 else li > 0) else 0.0) for r in range(2)) * 2 / 2
        want += [torch.full((16, 16), gw), torch.full((16,), gb)]
    for r in range(2):
        for got, w in zip(res[r][2], want):
            assert torch.allclose(torch.from_numpy(got), w, atol=1e-6)
    # step 1: every layer is used by some rank -> nothing hidden; step 2: layer 3 unused everywhere -> hidden on both
    # ranks; the probe of the captured form reports the unreached layer 3
    assert res[0][3] == [106, 107, 206, 207] and res[1][3] == [106, 107, 206, 207]
    # two-stage exchange: every layer's gradient is the mean over the ranks (x = 1 and 2 -> weight 3, bias 2)
    for r in range(2):
        for li in range(4):
            assert torch.allclose(torch.from_numpy(res[r][4][2 * li]), torch.full((16, 16), 3.0), atol=1e-6)
            assert torch.allclose(torch.from_numpy(res[r][4][2 * li + 1]), torch.full((16,), 2.0), atol=1e-6)


def _worker8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gvl_amd.parallel import GradBuckets, shard_batch
        # ---- an uneven batch: 13 videos over 8 ranks, two of them without any event ---------------------------------
        B = 13
        n_gt = [2, 0, 3, 1, 0, 2, 1, 4, 1, 2, 3, 1, 2]          # videos 1 and 4 (ranks 1 and 4) have no event
        dt = {"video_tensor": torch.arange(B)[:, None, None].float().expand(B, 4, 2), "video_mask": torch.ones(B, 4, dtype=torch.bool),
              "video_length": torch.zeros(B, 3), "gt_boxes_mask": torch.ones(B, 3, dtype=torch.bool),
              "video_target": [{"boxes": torch.zeros(n, 2), "labels": torch.zeros(n)} for n in n_gt],
              "cap_raw": [["c"] * n for n in n_gt], "cap_tensor": torch.arange(sum(n_gt))[:, None].expand(-1, 6),
              "cap_mask": torch.ones(sum(n_gt), 6)}
        mine = shard_batch(dt, rank, world)
        vids = mine["video_tensor"][:, 0, 0].long().tolist()
        assert vids == list(range(rank, B, world)) and mine["cap_tensor"].shape[0] == sum(n_gt[v] for v in vids)
        # num_boxes: the mean over the ranks of the local target count (criterion.py:178-181), >= 1
        nb = torch.tensor([float(sum(n_gt[v] for v in vids))])
        dist.all_reduce(nb)
        nb = (nb / world).clamp(min=1.0)
        # ---- gradient buckets at their real size: 25 MB buckets over a 34 MB parameter set, three steps ------------
        torch.manual_seed(0)
        layers = [torch.nn.Linear(1024, 1024) for _ in range(8)] + [torch.nn.Linear(1024, 520)]
        params = [p for l_ in layers for p in l_.parameters()]
        buckets = GradBuckets(params)                             # default bucket_bytes = 25 MB, overlap hooks
        sizes = [(e - s_) * 4 for s_, e, _ in buckets.buckets]
        assert len(sizes) == 2 and sizes[0] >= 25 << 20 and sum(sizes) == 4 * sum(p.numel() for p in params)
        x = torch.full((1, 1024), float(rank + 1))
        seq = []
        orig = dist.all_reduce

        def spy(t, *a, **k):
            seq.append(t.numel())
            return orig(t, *a, **k)
        dist.all_reduce = spy
        try:
            for step in range(3):
                buckets.zero()
                # every rank walks the layers in its own order and skips one (rank-dependent) layer: completion order of
                # the buckets differs between ranks, the posted order must not
                order = layers[rank % 9:] + layers[:rank % 9]
                used = [l_ for i, l_ in enumerate(order) if i != (rank + step) % 9]
                sum(l_(x).sum() for l_ in used).backward()
                buckets.finish()
        finally:
            dist.all_reduce = orig
        assert seq == [sizes[0] // 4, sizes[1] // 4] * 3, seq      # bucket 0 then bucket 1, every step, on every rank
        q.put((rank, float(nb), [float(p.grad.double().sum()) for p in params[:4]], len(buckets.unused_params())))
    except Exception:
        import traceback
        q.put((rank, "error", traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_uneven_batch_empty_ranks_and_bucket_order():
    """the data-parallel machinery at W = 8 (the node's GPU count) -- B not a multiple of W, ranks whose videos
    carry no event, 25 MB bucket boundaries, the same all-reduce sequence on every rank although their backward orders differ"""
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for r in res:
        assert r[1] != "error", r[2]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    n_gt = [2, 0, 3, 1, 0, 2, 1, 4, 1, 2, 3, 1, 2]
    assert all(abs(r[1] - sum(n_gt) / 8) < 1e-6 for r in res)       # 22 targets / 8 ranks = 2.75 on every rank
    # the gradients every rank holds after the exchange are identical (the mean over the ranks)
    for r in res[1:]:
        assert r[2] == res[0][2] and r[3] == res[0][3]


def _worker_agree(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gvl_amd.parallel import GradBuckets
        model = _make_model()
        params = list(model.parameters())                    # 6 tensors
        b = GradBuckets(params, bucket_bytes=1024, flat=True, overlap=False)
        # what each rank's probe found unused on ITS batch: parameter 0 everywhere; parameter 1 on the ranks with events except
        # rank 2 (which reaches it); parameters 4, 5 only on the ranks WITHOUT events (a batch without events skips the captioner)
        has_events = rank not in (1, 3)
        local = [params[0]] + ([params[1]] if rank != 2 else []) + ([] if has_events else [params[4], params[5]])
        got = b.agree_unused(local, has_events)
        res = None if got is None else [i for i, p in enumerate(params) if any(p is g for g in got)]
        # nobody has events: nothing can be concluded
        none = b.agree_unused(params, False)
        q.put((rank, "ok", (res, none)))
    except Exception as e:                                   # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc() + str(e)))
    finally:
        dist.destroy_process_group()


def test_structurally_unused_parameters_are_agreed_across_the_ranks():
    """the set of parameters a layout-keyed data-parallel capture hides from clip + Adam is agreed ONCE --
    unused on EVERY rank whose batch had events; ranks without events abstain; no rank with events -> no conclusion"""
    world = 4
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_agree, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r[2]
        assert r[2] == ([0], None), r                        # parameter 0 only; the same list on every rank
