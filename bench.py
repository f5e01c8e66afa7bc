#!/usr/bin/env python3
"""Contract benchmark: videos/sec of the GVL (PDVC) eval forward -- or train-step ms -- on N MI355X GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode eval|train] [--T 100] [--queries 300]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch of synthetic input: ``PDVC.forward(dt, criterion, None,
'queries', eval_mode=True)`` (eval_utils.py:203) on B=16 videos per GPU of the ANet-TSP model
(cfgs/anet_tsp_ssvg.yml with num_queries=300, T=100, random-init weights, synthetic features).
Videos are independent, so ranks shard by video with no data-path collective ("weak" scaling: 16 videos per GPU).
Rank 0 prints ONE JSON line; it carries the `roofline` object of the deformable-attention kernel (measured live with
HIP events on the launch stream) and, at N=1, the `cpu_baseline` object (the oracle's CPU restatement of the same
forward with the reference's CPU-fallback sampling semantics, timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_TBS = 8.0           # MI355X HBM3E peak (MI355X_MICROARCH.md)


def synth_batch(B, T, feat, vocab, n_gt, device, seed=1):
    """SURVEY.md section 8d synthetic `dt` (keys of pdvc.py:250-258 + targets for the criterion)."""
    g = torch.Generator().manual_seed(seed)
    vt = torch.randn(B, T, feat, generator=g)
    vmask = torch.ones(B, T, dtype=torch.bool)
    vlen = torch.tensor([[float(T), 120.0, float(n_gt)]] * B)
    targets = []
    for _ in range(B):
        c = torch.rand(n_gt, generator=g) * 0.5 + 0.25
        l_ = torch.rand(n_gt, generator=g) * 0.3 + 0.1
        targets.append({"boxes": torch.stack([c, l_], -1).to(device), "labels": torch.zeros(n_gt, dtype=torch.long,
                                                                                           device=device)})
    cap_len = 12
    caps = torch.randint(1, vocab, (B * n_gt, cap_len), generator=g)
    caps[:, 0] = 0
    caps[:, -1] = 0
    return {"video_tensor": vt.to(device), "video_mask": vmask.to(device), "video_length": vlen.to(device),
            "video_target": targets, "cap_raw": [["x"] * n_gt] * B, "cap_tensor": caps.to(device),
            "cap_mask": torch.ones(B * n_gt, cap_len, device=device),
            "gt_boxes_mask": torch.ones(B, n_gt, dtype=torch.bool, device=device)}


def msda_bytes(B, S, Q, M=8, L=4, P=4, C=512, value_bytes=4):
    """algorithmic bytes of one forward launch: value + loc(2) + weight + output; fp32, or
    bf16 value / output with fp32 locations and weights ("bf16 value/out halves the C terms")"""
    return B * (value_bytes * S * C + 4 * 3 * Q * M * L * P + value_bytes * Q * C)


def kernel_times(entries):
    """group the in-library dispatch timings (gvl_prof_collect) by (kernel, meta_a, meta_b) -> (mean us, count)"""
    acc = {}
    for tag, ma, mb, us in entries:
        acc.setdefault((tag, ma, mb), []).append(us)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def cfg_l_probe(dev, B, T=512, Q=300, M=8, L=4, P=4, iters=20):
    """The same kernel on the long-video launch shape (cfg L: T = 512, S = 960; level 0 read from global memory):
    `iters` fused decoder-shaped launches on synthetic operands, timed with the library's per-dispatch stamps.  A
    supplementary data point for DESIGN.md section 4.7 -- not part of the timed region, not part of `value`."""
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    from gvl_amd.deformable_transformer import make_level_tensors
    from gvl_amd.ops.modules.ms_deform_attn import temporal_shapes_2d
    lens = [T]
    for _ in range(L - 1):
        lens.append((lens[-1] - 1) // 2 + 1)
    S = sum(lens)
    tsh, lsi = make_level_tensors(lens, dev)
    shapes2d = temporal_shapes_2d(tsh, lsi)
    g = torch.Generator(device=dev).manual_seed(7)
    value = torch.randn(B, S, M, 64, device=dev, generator=g)
    proj = torch.randn(B, Q, 2 * M * L * P, device=dev, generator=g)
    ref = torch.rand(B, Q, L, 1, device=dev, generator=g)
    for _ in range(3):
        MSDA.msda1d_fused_forward(value, shapes2d, lsi, proj, ref, L, P)
    torch.cuda.synchronize()
    MSDA.profile_enable(True)
    for _ in range(iters):
        MSDA.msda1d_fused_forward(value, shapes2d, lsi, proj, ref, L, P)
    torch.cuda.synchronize()
    MSDA.profile_enable(False)
    us = [e[3] for e in MSDA.profile_collect()]
    us = sum(us) / len(us)
    nbytes = msda_bytes(B, S, Q)
    return {"T": T, "S": S, "kernel_us": round(us, 2), "launches_timed": iters, "algorithmic_bytes": nbytes,
            "frac": round(nbytes / (us * 1e-6) / 1e9 / (HBM_PEAK_TBS * 1e3), 4),
            "note": "back-to-back launches on synthetic operands after the timed region"}


FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32-input MFMA peak (no xf32 / TF32 on gfx950)


def gemm_probe(model, dev, rows, iters=20):
    """Where the eval forward's TIME goes is not the deformable-attention kernel but the captioner's library GEMMs
    (hipBLASLt through PyTorch; 68 % of the kernel time, profiles/r01_eval_kernel_stats.txt).  The largest of them,
    the vocabulary logits (rows x 512 x (V+1)) of every token step, is timed here with stream events around `iters`
    back-to-back calls after the timed region, and priced against the fp32 MFMA peak."""
    head = model.caption_head[-1]
    w, b = head.logit.weight, head.logit.bias
    x = torch.randn(rows, w.shape[1], device=dev)
    for _ in range(3):
        torch.nn.functional.linear(x, w, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.nn.functional.linear(x, w, b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    flops = 2.0 * rows * w.shape[0] * w.shape[1]
    tf = flops / (us * 1e-6) / 1e12
    return {"kernel": f"hipBLASLt fp32 GEMM {rows}x{w.shape[1]}x{w.shape[0]} (vocabulary logits, once per token step)",
            "bound": "mfma", "achieved": round(tf, 1), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 4), "kernel_us": round(us, 1), "calls_timed": iters,
            "note": "library kernel, not hand-written; back-to-back calls after the timed region (torch events on "
                    "the launch stream)"}


def cpu_baseline(model, opt, T, seconds_budget=25.0):
    """The oracle's CPU port (reference CPU-fallback semantics: grid_sample border) on a bounded sample."""
    from oracle import torch_ref as R
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    # a few hundred tiny ops per token step: more than ~32 threads only adds fork/join overhead on a big host
    ncores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(ncores)
    nvid = 2
    dt = synth_batch(nvid, T, opt.feature_dim, opt.vocab_size, 3, "cpu", seed=1)
    t0 = time.perf_counter()
    with torch.no_grad():
        R.pdvc_eval_forward(sd, dt, n_enc=opt.enc_layers, n_dec=opt.dec_layers, pad_mode="border",
                            max_caption_len=opt.max_caption_len)
    el = time.perf_counter() - t0
    done = nvid
    # one more, larger batch if the budget allows (amortises per-call overheads the way eval_batch_size=16 does)
    if el < seconds_budget / 5:
        nvid2 = min(16, max(2, int(nvid * (seconds_budget * 0.6) / max(el, 1e-3))))
        dt = synth_batch(nvid2, T, opt.feature_dim, opt.vocab_size, 3, "cpu", seed=1)
        t0 = time.perf_counter()
        with torch.no_grad():
            R.pdvc_eval_forward(sd, dt, n_enc=opt.enc_layers, n_dec=opt.dec_layers, pad_mode="border",
                                max_caption_len=opt.max_caption_len)
        el = time.perf_counter() - t0
        done = nvid2
    return {"value": done / el, "unit": "videos/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{done} synthetic videos, one eval forward incl. {opt.max_caption_len + 1} greedy caption "
                      f"steps, oracle/torch_ref.py (grid_sample border = reference CPU fallback), {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="eval", choices=["eval", "train"])
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--queries", type=int, default=300)
    ap.add_argument("--batch", type=int, default=16, help="videos per GPU")
    ap.add_argument("--cfg", default="anet_tsp_ssvg", help="gvl_amd.config.CONFIGS entry (BASELINE config 4: "
                    "--cfg yc2_tsn_dvc --T 512 --queries 100 --dtype bf16)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16 = torch.autocast(bfloat16): bf16 GEMMs + bf16-storage deformable attention, fp32 captioner")
    ap.add_argument("--no-captioner", action="store_true", help="eval_disable_captioning=True (diagnostic only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="do not replay the caption decoding loop from a hipGraph")
    ap.add_argument("--split-exchange", action="store_true",
                    help="--mode train on one GPU in the data-parallel form (two graphs + eager exchange point); diagnostic")
    ap.add_argument("--no-tuned-gemm", action="store_true",
                    help="keep hipBLASLt's default kernel choice instead of gvl_amd/tunableop_mi355x.csv")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {a.gpus}")
    # test knobs (tests/test_gpu_bench_dp.py): several ranks on ONE GPU over gloo exercise the N > 1 control flow
    dev_index = int(os.environ.get("GVL_BENCH_DEVICE", local_rank))
    backend = os.environ.get("GVL_DIST_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # a checkout without the (git-ignored) built library: compile it once, on one rank, before anyone loads it
    from gvl_amd import build as lib_build
    if not os.path.exists(lib_build.OUT):
        if local_rank == 0:
            lib_build.build()
        if world > 1:
            dist.barrier()
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    from gvl_amd.tuning import enable_tuned_gemms
    tuned = (not a.no_tuned_gemm) and enable_tuned_gemms()
    opt = make_opt(a.cfg, num_queries=a.queries, frame_embedding_num=a.T,
                   eval_disable_captioning=bool(a.no_captioner), device="cuda")
    torch.manual_seed(0)
    model, criterion, _, _ = build(opt)
    model = model.to(dev)
    B = a.batch
    dt = synth_batch(B, a.T, opt.feature_dim, opt.vocab_size, 3, dev, seed=1 + rank)

    from gvl_amd import MultiScaleDeformableAttention as MSDA

    if a.mode == "eval":
        model.eval()
        for head in model.caption_head:
            head.graph_decode = not a.no_graph

        if a.no_graph:
            def step():
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.dtype == "bf16"):
                    return model(dt, criterion, None, "queries", eval_mode=True)
        else:
            # the whole forward (not only the decoding loop) replayed from one hipGraph; the kernel stamps of the
            # roofline block come from instrumented eager forwards after the timed region (as in --mode train)
            from gvl_amd.parallel import GraphedEvalForward
            graphed_eval = GraphedEvalForward(model, criterion,
                                              autocast_dtype=torch.bfloat16 if a.dtype == "bf16" else None)

            def step():
                return graphed_eval(dt)
    else:
        from gvl_amd.parallel import GraphedTrainStep, TrainStep
        model.train()
        # one process: the whole step is ONE hipGraph; several processes: forward/backward graph, eager bucketed RCCL
        # all-reduce of the flat gradient buffer, clip/Adam graph (no collective inside a graph) -- GraphedTrainStep
        use_graph = not a.no_graph
        ac = torch.bfloat16 if a.dtype == "bf16" else None
        if use_graph:
            trainer = GraphedTrainStep(model, criterion, opt, world_size=world,
                                       split_exchange=True if a.split_exchange else None, autocast_dtype=ac)
        else:
            trainer = TrainStep(model, criterion, opt, world_size=world, autocast_dtype=ac)

        def step():
            return trainer(dt)

    step()            # set-up, not a warm-up step: one-time work (hipGraph capture, library handles, lazily built constants)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    MSDA.profile_enable(True)          # per-dispatch begin/end stamps of the library's kernels (hipExtLaunchKernel)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    MSDA.profile_enable(False)
    ktimes = kernel_times(MSDA.profile_collect())
    roofline_source = "per-dispatch stamps (hipExtLaunchKernel events) of the launches inside the timed region"
    if not a.no_graph:
        # the timed steps are hipGraph replays: the library launches nothing at replay time, so the kernel stamps come
        # from two instrumented eager steps run right after the timed region (same process, same inputs)
        MSDA.profile_enable(True)
        for _ in range(2):
            if a.mode == "train":
                trainer._eager(dt)
            else:
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.dtype == "bf16"):
                    model(dt, criterion, None, "queries", eval_mode=True)
        torch.cuda.synchronize()
        MSDA.profile_enable(False)
        ktimes = kernel_times(MSDA.profile_collect())
        roofline_source = "two instrumented eager steps right after the timed region (timed steps are hipGraph replays)"
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    ms_per_step = elapsed * 1e3 / a.steps
    videos_per_s = world * B * a.steps / elapsed

    # roofline of the deformable-attention forward kernel (decoder cross-attention launch shape = the larger one)
    lens = [a.T]
    for _ in range(opt.num_feature_levels - 1):
        lens.append((lens[-1] - 1) // 2 + 1)
    S = sum(lens)
    roof = None
    fwd = {k: v for k, v in ktimes.items() if k[0] in ("fwd_t1d_d64", "fwd_generic")}
    dec_key = next((k for k in fwd if k[1] == a.queries and k[2] == B), None)
    if dec_key is not None:
        us, n = fwd[dec_key]
        vb = 2 if a.dtype == "bf16" else 4
        nbytes = msda_bytes(B, S, a.queries, value_bytes=vb)
        achieved = nbytes / (us * 1e-6) / 1e9              # GB/s
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc) and (B, a.T, a.queries, a.dtype) == (16, 100, 300, "f32"):
            # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same launch shape (FETCH_SIZE and
            # WRITE_SIZE in separate passes, gfx950 2x FETCH correction); counters cannot be read from inside the run
            traffic = json.load(open(pmc))["k_fwd_t1d_d64_fused_dec"]["hbm_bytes_corrected"]   # the variant the model launches
            traffic_src = "profiles/r01_pmc_traffic.json"
        roof = {"bound": "hbm", "kernel": f"k_{dec_key[0]} (decoder cross-attention launch, Lq={a.queries})",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s",
                "frac": round(achieved / (HBM_PEAK_TBS * 1e3), 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel_us": round(us, 2), "launches_timed": n, "algorithmic_bytes": nbytes,
                "source": roofline_source}
        enc_key = next((k for k in fwd if k[1] == S and k[2] == B), None)
        if enc_key is not None:
            eus, en = fwd[enc_key]
            eb = msda_bytes(B, S, S, value_bytes=vb)
            roof["encoder_launch"] = {"kernel_us": round(eus, 2), "launches_timed": en, "algorithmic_bytes": eb,
                                      "frac": round(eb / (eus * 1e-6) / 1e9 / (HBM_PEAK_TBS * 1e3), 4)}
    other = {f"{k[0]}[{k[1]}]": {"us": round(v[0], 2), "n": v[1]} for k, v in ktimes.items() if k not in fwd}
    if roof is not None and rank == 0 and a.T != 512 and a.dtype == "f32":
        roof["cfg_L_launch"] = cfg_l_probe(dev, B)

    line = {
        "metric": "videos/sec (eval fwd)" if a.mode == "eval" else "train-step ms",
        "value": round(videos_per_s, 3) if a.mode == "eval" else round(ms_per_step, 3),
        "unit": "videos/s" if a.mode == "eval" else "ms",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": a.mode == "eval", "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if a.dtype == "f32" else "bf16 storage + bf16 GEMMs, f32 accumulate / locations / captioner",
        "data": "synthetic",
        "config": {"workload": f"cfgs/{a.cfg}.yml PDVC {'eval forward' if a.mode == 'eval' else 'train step'}"
                               f" B={B}/GPU T={a.T} L=4 Q={a.queries}, "
                               + ("captioner off (diagnostic)" if a.no_captioner else
                                  f"LSTM-DSA greedy captioning of {opt.max_caption_len} tokens ({opt.max_caption_len} token steps; the "
                                  f"reference's extra step after the last token is dead code -- never read -- and is "
                                  f"not evaluated)")
                               + ", set criterion + Hungarian matcher on 3 GT/video",
                   "library_gemm_selection": "gvl_amd/tunableop_mi355x.csv" if tuned else "hipBLASLt default",
                   "global_batch": world * B, "parallelism": f"dp{world} (videos sharded, no data-path collective)"
                   if a.mode == "eval" else f"dp{world} (RCCL gradient all-reduce)"},
        "roofline": roof,
        "kernels_us": other,
    }
    if a.mode == "train":
        line["videos_per_s"] = round(videos_per_s, 3)
    if rank == 0 and a.mode == "eval" and not a.no_captioner and a.dtype == "f32":
        with torch.no_grad():
            line["dominant_library_gemm"] = gemm_probe(model, dev, B * a.queries)
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.mode == "eval":
        line["cpu_baseline"] = cpu_baseline(model, opt, a.T)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()                    # rank 0 ran the supplementary probes; leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
