#!/usr/bin/env python3
"""Contract benchmark: videos/sec of the GVL (PDVC) eval forward AND train-step ms on N MI355X GPUs, in one JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode both|eval|train] [--T 100] [--queries 300]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch of synthetic input: ``PDVC.forward(dt, criterion, None,
'queries', eval_mode=True)`` (eval_utils.py:203) -- or, for the second half of BASELINE.json's metric, one training
step (train.py:385-409: forward, Hungarian matching, losses, backward, clip, Adam) -- on B=16 videos per GPU of the
ANet-TSP model (cfgs/anet_tsp_ssvg.yml with num_queries=300, T=100, random-init weights, synthetic features;
SURVEY.md section 8d).  The timed region ROTATES through `--rotate` (default 8) different batches -- 0..10 events per
video, different caption lengths and caption-tensor widths, as eval_utils.py:187-203 / train.py:385-398 feed them -- so
that the captured hipGraphs are exercised the way a real loop exercises them (`graphs` in the line reports how many
graphs exist and that none was captured inside the timed region).  `--fixed-layout` restores round 1's single
3-events-per-video batch.
Videos are independent, so ranks shard by video with no data-path collective in eval ("weak" scaling: 16 videos per
GPU); the train step exchanges gradients over RCCL.
Rank 0 prints ONE JSON line; it carries `value` (eval videos/s), `train_step_ms`, the `roofline` object of the
deformable-attention forward kernel and `train_roofline` of the backward kernel (both measured live with HIP events
on the launch stream), and, at N=1, the `cpu_baseline` object (the oracle's CPU restatement of the same forward with the
reference's CPU-fallback sampling semantics, timed on this host's cores as BASELINE.md section 3 prescribes, on a
bounded sample).
"""
import argparse
import json
import os
import statistics
import sys
import time

# before the first HIP call of the process (see gvl_amd/__init__.py): hipGraph replays without the runtime's packet-capture
# fast path, which mis-orders the non-kernel nodes of a captured step after a host synchronisation on ROCm 7.2
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_TBS = 8.0           # MI355X HBM3E peak (MI355X_MICROARCH.md)
PMC_FILE = os.path.join("profiles", "r06_pmc_traffic.json")


def synth_batch(B, T, feat, vocab, n_gt, device, seed=1, cap_words=10):
    """SURVEY.md section 8d synthetic `dt` (keys of pdvc.py:250-258 + targets for the criterion).
    n_gt: events per video, an int (every video) or a list (per video, 0 allowed).  cap_words: words per caption, an
    int or a (lo, hi) range drawn per caption; rows are <bos>=0, words, <eos>=0, <pad>=0 and the tensor is as wide as
    the longest caption + 2, exactly what the reference's collate_fn builds (video_dataset.py:68-88)."""
    g = torch.Generator().manual_seed(seed)
    ns = [int(n_gt)] * B if isinstance(n_gt, int) else [int(n) for n in n_gt]
    vt = torch.randn(B, T, feat, generator=g)
    vmask = torch.ones(B, T, dtype=torch.bool)
    vlen = torch.tensor([[float(T), 120.0, float(n)] for n in ns])
    targets = []
    for n in ns:
        c = torch.rand(n, generator=g) * 0.5 + 0.25
        l_ = torch.rand(n, generator=g) * 0.3 + 0.1
        targets.append({"boxes": torch.stack([c, l_], -1).to(device), "labels": torch.zeros(n, dtype=torch.long,
                                                                                           device=device)})
    total = sum(ns)
    if isinstance(cap_words, int):
        words = [cap_words] * total
    else:
        words = torch.randint(cap_words[0], cap_words[1] + 1, (total,), generator=g).tolist()
    width = max(words + [1]) + 2
    caps = torch.zeros(total, width, dtype=torch.long)
    cmask = torch.zeros(total, width)
    for i, w in enumerate(words):
        caps[i, 1:1 + w] = torch.randint(1, vocab, (w,), generator=g)
        cmask[i, :w + 2] = 1
    mx = max(ns + [1])
    return {"video_tensor": vt.to(device), "video_mask": vmask.to(device), "video_length": vlen.to(device),
            "video_target": targets, "cap_raw": [["x"] * n for n in ns], "cap_tensor": caps.to(device),
            "cap_mask": cmask.to(device),
            "gt_boxes_mask": torch.tensor([[k < n for k in range(mx)] for n in ns], dtype=torch.bool, device=device)}


def in_graph_kernel_us(a, mode):
    """Durations of the deformable-attention launches inside the REPLAYED hipGraphs: a child `rocprofv3 --kernel-trace -- python3
    bench.py --no-probes ...` of the same workload (20 timed replays), its CSV read per launch position -- the four forward
    launches of a step come encoder, encoder, decoder, decoder, the backward's in the reverse order; means over the second half of
    the run (= the timed replays).  -> {"fwd": {...}, "bwd": {...}} or {"error": ...}; never raises (a box without rocprofv3, or
    a profiler that fails, leaves the eager stamps as the only source and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    out = tempfile.mkdtemp(prefix="gvl_ingraph_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--mode", mode, "--no-cpu-baseline", "--no-probes", "--steps", "20", "--warmup", "5", "--batch", str(a.batch),
           "--T", str(a.T), "--queries", str(a.queries), "--cfg", a.cfg, "--rotate", str(a.rotate)]
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
        files = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
        if p.returncode != 0 or not files:
            return {"error": f"rocprofv3 child failed (rc {p.returncode}): {p.stderr[-300:]}"}
        rows = {"fwd": [], "bwd": []}
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                n = r["Kernel_Name"]
                if "k_fwd_t1d_d64" in n:
                    rows["fwd"].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
                elif "k_bwd_t1d_split" in n or "k_bwd_t1d_own" in n or "k_bwd_t1d_d64" in n:
                    rows["bwd"].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        # the EVAL forward's launches (what `value` is made of): groups of four forward launches that no backward launch follows
        # before the next group -- a train step's four are followed by its backward's (both steps launch the same instantiation
        # since the training forward also leaves the output rows' maxima)
        fw, bw_starts = sorted(rows["fwd"]), sorted(t0 for t0, _ in rows["bwd"])
        if len(fw) % 4 == 0 and bw_starts:
            import bisect
            groups = [fw[i:i + 4] for i in range(0, len(fw), 4)]
            eval_groups = []
            for gi, grp in enumerate(groups):
                nxt = groups[gi + 1][0][0] if gi + 1 < len(groups) else float("inf")
                j = bisect.bisect_right(bw_starts, grp[-1][1])
                if not (j < len(bw_starts) and bw_starts[j] < nxt):
                    eval_groups += grp
            if eval_groups:
                rows["fwd"] = eval_groups
        res = {}
        for kind, dec_pos in (("fwd", (2, 3)), ("bwd", (0, 1))):
            rs = sorted(rows[kind])
            if len(rs) < 16 or len(rs) % 4:
                continue
            pos = [[], [], [], []]
            for i, (t0, t1) in enumerate(rs):
                pos[i % 4].append((t1 - t0) / 1e3)
            tail = [v[len(v) // 2:] for v in pos]
            mean = [sum(v) / len(v) for v in tail]
            res[kind] = {"decoder_us": round((mean[dec_pos[0]] + mean[dec_pos[1]]) / 2, 2),
                         "encoder_us": round(sum(mean[i] for i in range(4) if i not in dec_pos) / 2, 2),
                         "per_position_us": [round(m, 2) for m in mean], "launches": len(rs),
                         "source": "rocprofv3 --kernel-trace of a child run of this script (20 timed replays), means over the "
                                   "second half of the run"}
        return res or {"error": "no deformable-attention launches in the trace"}
    except Exception as e:                                             # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(out, ignore_errors=True)


def rotating_batches(R, B, T, feat, vocab, device, seed):
    """R batches whose layouts all differ: 0..10 events per video, caption lengths 3..(8..20) words."""
    g = torch.Generator().manual_seed(1000 + seed)
    out = []
    for r in range(R):
        ns = torch.randint(0, 11, (B,), generator=g).tolist()
        if r == 0:
            ns[0], ns[1] = 0, 10                      # the extremes are always present
        hi = 8 + (5 * r) % 13
        out.append(synth_batch(B, T, feat, vocab, ns, device, seed=seed * 100 + r, cap_words=(3, hi)))
    return out


def msda_bytes(B, S, Q, M=8, L=4, P=4, C=512, value_bytes=4):
    """algorithmic bytes of one forward launch: value + loc(2) + weight + output; fp32, or
    bf16 value / output with fp32 locations and weights ("bf16 value/out halves the C terms")"""
    return B * (value_bytes * S * C + 4 * 3 * Q * M * L * P + value_bytes * Q * C)


def msda_bwd_bytes(B, S, Q, M=8, L=4, P=4, C=512, value_bytes=4):
    """backward launch: read value, loc, w, grad_out; write grad_value, grad_loc, grad_w (zero fill excluded)"""
    return B * (2 * value_bytes * S * C + 4 * 6 * Q * M * L * P + value_bytes * Q * C)


def kernel_times(entries):
    """group the in-library dispatch timings (gvl_prof_collect) by (kernel, meta_a, meta_b) -> (median us, count, mean us).
    The MEDIAN is what the roofline blocks use: of a dozen event-stamped eager launches of one kernel one or two read
    1.3-2x the others (tools/fwd_insitu_samples.py: 9.8 10.1 9.7 10.0 19.0 15.4 9.9 ... us) -- an artefact of stamping
    eager launches (the rocprofv3 trace of the replayed graph, profiles/r02_msda_launches_in_graph.txt, shows a mean within
    5 % of its minimum) -- and with a handful of samples such a reading moves the mean by 20 %."""
    acc = {}
    for tag, ma, mb, us in entries:
        acc.setdefault((tag, ma, mb), []).append(us)
    out = {}
    for k, v in acc.items():
        w = sorted(v)
        med = w[len(w) // 2] if len(w) % 2 else 0.5 * (w[len(w) // 2 - 1] + w[len(w) // 2])
        out[k] = (med, len(v), sum(v) / len(v))
    return out


def kernel_probe(dev, B, T=512, Q=300, M=8, L=4, P=4, iters=20, backward=False):
    """The same kernels at other launch sizes -- cfg L (T = 512, S = 960; level 0 read from global memory) and the
    latency-free batch B = 64: `iters` fused decoder-shaped launches on synthetic operands, timed with the library's
    per-dispatch stamps.  Supplementary data points (at cfg A one launch moves
    23-37 MB, i.e. 3-5 us of HBM time against a ~5 us floor of launch ramp + one staging round trip + the VALU-bound
    sample loop) -- not part of the timed region, not part of `value`."""
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
    gout = torch.randn(B, Q, M * 64, device=dev, generator=g) if backward else None

    def call():
        if backward:
            MSDA.msda1d_fused_backward(value, shapes2d, lsi, proj, ref, gout, L, P)
        else:
            MSDA.msda1d_fused_forward(value, shapes2d, lsi, proj, ref, L, P)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    MSDA.profile_enable(True)
    for _ in range(iters):
        call()
    torch.cuda.synchronize()
    MSDA.profile_enable(False)
    us = sum(e[3] for e in MSDA.profile_collect()) / iters              # backward: + k_sum_partials where it still runs
    nbytes = (msda_bwd_bytes if backward else msda_bytes)(B, S, Q)
    return {"B": B, "T": T, "S": S, "kernel_us": round(us, 2), "launches_timed": iters, "algorithmic_bytes": nbytes,
            "frac": round(nbytes / (us * 1e-6) / 1e9 / (HBM_PEAK_TBS * 1e3), 4),
            "note": "back-to-back launches on synthetic operands after the timed region"}


F16_MFMA_PEAK_TFLOPS = 2500.0          # dense fp16 / bf16 MFMA (MI355X_MICROARCH.md)
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


def cpu_baseline(model, opt, T, budget_s=30.0):
    """the oracle's CPU port (reference CPU-fallback semantics: grid_sample border) on this
    host's cores, 2 warm-up iterations and the median of >= 5 timed ones, eval forward WITH and WITHOUT the captioner.
    Threads: BASELINE.md prescribes torch.set_num_threads(os.cpu_count()); on the 256-thread GPU hosts that setting is
    pathological for this workload (a few hundred small ops per token step: 42 s per VIDEO, 60x slower than 32
    threads, measured in round 2), so a short probe (2 videos, captioner off) times both os.cpu_count() and 32 threads
    and the protocol runs with the FASTER of the two -- the CPU gets its best configuration; both probe times and the
    thread count used are reported.  Bounded sample: the batch with the captioner is sized from one probe iteration so
    that its 7 iterations fit `budget_s`; the batch actually used is stated in `sample`."""
    from oracle import torch_ref as R
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ncpu = os.cpu_count() or 1

    def run(nvid, captioning):
        dt = synth_batch(nvid, T, opt.feature_dim, opt.vocab_size, 3, "cpu", seed=1)
        t0 = time.perf_counter()
        with torch.no_grad():
            R.pdvc_eval_forward(sd, dt, n_enc=opt.enc_layers, n_dec=opt.dec_layers, pad_mode="border",
                                max_caption_len=opt.max_caption_len, captioning=captioning)
        return time.perf_counter() - t0

    def median_of(nvid, captioning, warm=2, timed=5):
        for _ in range(warm):
            run(nvid, captioning)
        ts = [run(nvid, captioning) for _ in range(timed)]
        return statistics.median(ts), ts

    probes = {}
    for n in sorted({ncpu, min(32, ncpu)}):
        torch.set_num_threads(n)
        run(2, False)
        probes[n] = run(2, False)
    threads = min(probes, key=probes.get)
    torch.set_num_threads(threads)
    # BASELINE.md section 3 times 16 videos per iteration; the sample is only cut below that if 7 iterations of 16 videos
    # would not fit the budget.  Sized from a WARMED 2-video run (the first call builds constants and is several times slower;
    # sizing from it used to cut the sample to 10 videos on hosts that run 16 in a second)
    run(2, True)
    probe = run(2, True) / 2.0
    nvid = max(1, min(16, int(budget_s / 7.0 / max(probe, 1e-3))))
    med, ts = median_of(nvid, True)
    nvid_nc = 16
    med_nc, ts_nc = median_of(nvid_nc, False)

    # train step (forward + Hungarian matcher + losses + backward) through the oracle's training
    # restatement (oracle/torch_ref.py:pdvc_train_forward, pinned to the reference by tests/test_oracle_golden.py)
    def run_train(nv):
        dt = synth_batch(nv, T, opt.feature_dim, opt.vocab_size, 3, "cpu", seed=1)
        leaf = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
        t0 = time.perf_counter()
        _, total = R.pdvc_train_forward(leaf, dt, n_enc=opt.enc_layers, n_dec=opt.dec_layers, pad_mode="border")
        total.backward()
        return time.perf_counter() - t0
    probe_t = run_train(2)
    nv_t = max(2, min(16, int(2 * budget_s * 0.4 / 7.0 / max(probe_t, 1e-3))))
    for _ in range(2):
        run_train(nv_t)
    tt = [run_train(nv_t) for _ in range(5)]
    med_t = statistics.median(tt)
    return {"value": nvid / med, "unit": "videos/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{nvid} synthetic videos per iteration (3 events each), eval forward incl. {opt.max_caption_len + 1} "
                      f"greedy caption steps (the reference's own count: its dead extra LSTM step after the last token is evaluated "
                      f"here and skipped by the GPU path -- about 3 % of this leg's time), oracle/torch_ref.py (grid_sample border = "
                      f"reference CPU fallback), "
                      f"2 warm-ups + median of 5: {med:.2f} s (runs {', '.join(f'{x:.2f}' for x in ts)})",
            "without_captioner": {"value": nvid_nc / med_nc, "unit": "videos/s",
                                  "sample": f"{nvid_nc} videos per iteration, eval_disable_captioning, 2 warm-ups + "
                                            f"median of 5: {med_nc:.2f} s"},
            "host_cpu_count": ncpu,
            "thread_probe_s": {str(k): round(v, 3) for k, v in probes.items()},
            "thread_probe_note": "seconds for one warmed 2-video forward without captioner at each thread count; the "
                                 "protocol above ran with the faster one (BASELINE.md section 3 names os.cpu_count())",
            "train_step": {"value": med_t * 1e3, "unit": "ms", "videos_per_step": nv_t,
                           "videos_per_s": nv_t / med_t,
                           "sample": f"forward + Hungarian matcher (scipy) + losses + autograd backward on {nv_t} videos "
                                     f"(3 events, 10-word captions), no optimizer step; 2 warm-ups + median of 5 "
                                     f"(runs {', '.join(f'{x:.2f}' for x in tt)} s)"}}


def timed_loop(step, batches, steps, warmup, world, dev):
    """W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize; -> elapsed seconds (max over
    ranks), per-rank seconds"""
    for i in range(warmup):
        o_ = step(batches[i % len(batches)])
        if os.environ.get("GVL_BENCH_TRACE_LOSS") and isinstance(o_, tuple) and isinstance(o_[0], torch.Tensor) and o_[0].numel() == 1:
            print(f"[trace] warm-up step {i}: loss {float(o_[0]):.4f}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    trace = bool(os.environ.get("GVL_BENCH_TRACE_LOSS"))       # dev: print every timed step's loss (synchronises: not a timing run)
    for i in range(steps):
        if trace and os.environ.get("GVL_BENCH_TRACE_GRADS") == str(i) and hasattr(step, "_forward_loss"):
            # dev: the gradients of THIS step's batch from an eager forward / backward on the current parameters (no update)
            mdl = step.model
            for p_ in mdl.parameters():
                p_.grad = None
            final, _l = step._forward_loss(batches[i % len(batches)])
            final.backward()
            gb = [(n, int((~torch.isfinite(p_.grad)).sum())) for n, p_ in mdl.named_parameters()
                  if p_.grad is not None and not bool(torch.isfinite(p_.grad).all())]
            gn = sorted(((float(p_.grad.float().norm()), n) for n, p_ in mdl.named_parameters() if p_.grad is not None), reverse=True)[:5]
            print(f"[trace] eager gradients before timed step {i}: loss {float(final):.4f}; non-finite: {gb[:10]}; largest norms {gn}",
                  file=sys.stderr, flush=True)
            for p_ in mdl.parameters():
                p_.grad = None
        out = step(batches[i % len(batches)])
        if trace and isinstance(out, tuple) and isinstance(out[0], torch.Tensor) and out[0].numel() == 1:
            print(f"[trace] timed step {i}: loss {float(out[0]):.4f}", file=sys.stderr, flush=True)
            if not bool(torch.isfinite(out[0])) and not getattr(timed_loop, "reported", False):
                timed_loop.reported = True
                bad = {k: float(v) for k, v in out[1].items() if isinstance(v, torch.Tensor) and v.numel() == 1
                       and not bool(torch.isfinite(v).all()) and "self_iou" not in k}
                print(f"[trace]   non-finite loss terms (besides self_iou): {bad}", file=sys.stderr, flush=True)
                mdl = getattr(step, "model", None)
                if mdl is not None:
                    gb = [n for n, p_ in mdl.named_parameters() if p_.grad is not None and not bool(torch.isfinite(p_.grad).all())]
                    pb = [n for n, p_ in mdl.named_parameters() if not bool(torch.isfinite(p_).all())]
                    print(f"[trace]   non-finite grads {len(gb)} {gb[:8]}; params {len(pb)} {pb[:3]}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    timed_loop.last_out = out                     # (read by the caller after the region: is the last step's loss finite?)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    try:                                   # the shader clock right behind the timed steps (diagnostic; outside the timing)
        from gvl_amd import MultiScaleDeformableAttention as _M
        timed_loop.last_clock_mhz = round(_M.clock_probe_mhz(dev, 4000), 0)
    except Exception:                      # noqa: BLE001
        timed_loop.last_clock_mhz = None
    per_rank = [elapsed]
    if world > 1:
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(x.item()) for x in every]
        elapsed = max(per_rank)
    return elapsed, per_rank


def self_launch(n, argv, timeout_s):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks the way the driver does --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <the
    same arguments>` -- as a CHILD job in its own process group, relay its stdout (rank 0's one JSON line) and return its
    exit code.  The caller has not initialised the GPU and nothing here does (no exec of a process that has: the ranks
    are fresh interpreters); a rank that fails makes the launcher exit non-zero, which is returned; a job that outlives
    `timeout_s` is killed as a group and reported as 124; exit 0 without a JSON line is reported as 1."""
    import signal
    import socket
    import subprocess
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, start_new_session=True)
    lines = []

    def relay():
        for ln in proc.stdout:
            if ln.startswith("{"):
                lines.append(ln)
            sys.stdout.write(ln)
            sys.stdout.flush()
    t = threading.Thread(target=relay, daemon=True)
    t.start()
    try:
        rc = proc.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.wait()
        print(f"bench.py: the {n} self-launched ranks did not finish within {timeout_s:.0f} s; killed", file=sys.stderr)
        return 124
    except KeyboardInterrupt:
        os.killpg(proc.pid, signal.SIGTERM)
        proc.wait()
        return 130
    t.join(timeout=10)
    if rc == 0 and len(lines) != 1:
        print(f"bench.py: the self-launched job printed {len(lines)} JSON lines, expected 1", file=sys.stderr)
        return 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="both", choices=["both", "eval", "train"])
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--queries", type=int, default=300)
    ap.add_argument("--batch", type=int, default=16, help="videos per GPU")
    ap.add_argument("--rotate", type=int, default=8, help="number of different batch layouts cycled through the timed region")
    ap.add_argument("--fixed-layout", action="store_true", help="one batch, 3 events per video, 10-word captions (round 1's workload)")
    ap.add_argument("--cfg", default="anet_tsp_ssvg", help="gvl_amd.config.CONFIGS entry (BASELINE config 4: "
                    "--cfg yc2_tsn_dvc --T 512 --queries 100 --dtype bf16)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16 = the step and the forward under torch.autocast(bfloat16); what that runs on: gvl_amd.pdvc.autocast_training_policy / "
                         "autocast_inference_policy")
    ap.add_argument("--no-captioner", action="store_true", help="eval_disable_captioning=True (diagnostic only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probes", action="store_true", help="skip the supplementary kernel / GEMM probes after the timed "
                    "regions (for rocprofv3 runs: the probes launch the same kernels at other sizes)")
    ap.add_argument("--no-graph", action="store_true", help="eager steps (no hipGraph replay)")
    ap.add_argument("--decode-chunk", type=int, default=5, help="tokens per captured decode segment (0: one graph)")
    ap.add_argument("--cap-len-policy", default="bucket", choices=["bucket", "grow"],
                    help="train graphs: one per caption-width bucket of 4 tokens (default: a batch pays the steps of its own "
                         "longest caption) or ONE at the widest caption width seen")
    ap.add_argument("--split-exchange", action="store_true",
                    help="train on one GPU in the data-parallel form (three graphs + eager exchange points); diagnostic")
    ap.add_argument("--no-tuned-gemm", action="store_true",
                    help="keep hipBLASLt's default kernel choice instead of gvl_amd/tunableop_mi355x.csv")
    ap.add_argument("--launch-timeout", type=float, default=3000.0,
                    help="seconds the self-launched ranks of `python bench.py --gpus N` (N > 1, no launcher) may take")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process has not touched the GPU (no HIP call so far, and
        # none below) -- it starts the ranks as a fresh child job and relays rank 0's line
        raise SystemExit(self_launch(a.gpus, sys.argv[1:], a.launch_timeout))
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

    # a checkout without the (git-ignored) built library, or with stale sources: rank 0 decides and builds, every rank
    # waits at the same unconditional barrier before anyone loads the library
    from gvl_amd import build as lib_build
    if rank == 0 and lib_build.needs_build():
        lib_build.build()
    if world > 1:
        dist.barrier()
    from gvl_amd.config import make_opt
    from gvl_amd.pdvc import build
    from gvl_amd.tuning import enable_tuned_gemms
    from gvl_amd.parallel import GraphedEvalForward, GraphedTrainStep, TrainStep
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    tuned = (not a.no_tuned_gemm) and enable_tuned_gemms()
    opt = make_opt(a.cfg, num_queries=a.queries, frame_embedding_num=a.T,
                   eval_disable_captioning=bool(a.no_captioner), device="cuda")
    torch.manual_seed(0)
    model, criterion, _, _ = build(opt)
    model = model.to(dev)
    B = a.batch
    if a.fixed_layout:
        batches = [synth_batch(B, a.T, opt.feature_dim, opt.vocab_size, 3, dev, seed=1 + rank)]
    else:
        batches = rotating_batches(max(1, a.rotate), B, a.T, opt.feature_dim, opt.vocab_size, dev, seed=1 + rank)
    ac = torch.bfloat16 if a.dtype == "bf16" else None
    lens = [a.T]
    for _ in range(opt.num_feature_levels - 1):
        lens.append((lens[-1] - 1) // 2 + 1)
    S = sum(lens)
    vb = 2 if a.dtype == "bf16" else 4
    from gvl_amd.pdvc import autocast_inference_policy as _aip
    vb_eval = 4 if (a.dtype == "f32" or _aip() != "bf16") else 2      # (autocast inference on the fp32-storage kernels)
    res = {}

    def traffic_of(cfg_key, launch):
        """HBM bytes per launch from the committed rocprofv3 --pmc passes (tools/pmc_run.sh: FETCH_SIZE and WRITE_SIZE in
        separate passes, gfx950 2x FETCH correction) of this same launch shape; counters cannot be read from inside the run"""
        pmc = os.path.join(ROOT, PMC_FILE)
        if os.path.exists(pmc) and (B, a.T, a.queries, a.dtype) == (16, 100, 300, "f32"):
            table = json.load(open(pmc))
            rec = table.get(cfg_key, {}).get(launch) if isinstance(table.get(cfg_key), dict) else None
            import hashlib
            src = os.path.join(ROOT, "gvl_amd", "csrc", "gvl_msda.hip")
            sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
            if rec and table.get("kernel_source_sha16") == sha:
                return rec["hbm_bytes_corrected"], PMC_FILE
            if rec:                     # counters were collected on other kernel sources: do not report them as current
                return None, f"{PMC_FILE} is stale (collected for kernel source {table.get('kernel_source_sha16')}, now {sha})"
        return None, None

    def instrumented(fn, n=12):
        """kernel stamps of `n` eager steps run right after a timed region whose steps were graph replays (the library
        launches nothing at replay time)"""
        MSDA.profile_enable(not os.environ.get("GVL_BENCH_NO_STAMPS"))     # (dev switch: the same eager steps, launched plainly)
        for i in range(n):
            fn(batches[i % len(batches)])
        torch.cuda.synchronize()
        MSDA.profile_enable(False)
        return kernel_times(MSDA.profile_collect())

    # ---------------------------------------------------------------------------------------------- eval half
    if a.mode in ("both", "eval"):
        model.eval()
        for head in model.caption_head:
            head.graph_decode = False

        def eager_eval(dt):
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.dtype == "bf16"):
                return model(dt, criterion, None, "queries", eval_mode=True)
        if a.no_graph:
            graphed_eval, step = None, eager_eval
        else:
            graphed_eval = GraphedEvalForward(model, criterion, autocast_dtype=ac, decode_chunk=a.decode_chunk)
            step = graphed_eval
        for dt in batches:          # set-up, not warm-up: one-time work (hipGraph capture at the capacity every layout
            step(dt)                # fits, library handles, lazily built constants)
        caps0 = graphed_eval.captures if graphed_eval else 0
        MSDA.profile_enable(a.no_graph)
        elapsed, per_rank = timed_loop(step, batches, a.steps, a.warmup, world, dev)
        eval_clock = getattr(timed_loop, "last_clock_mhz", None)
        MSDA.profile_enable(False)
        ktimes = kernel_times(MSDA.profile_collect()) if a.no_graph else instrumented(eager_eval)
        # the hand-written projection GEMM in front of every sampling launch: stamped in a pass of its own (level 2)
        MSDA.profile_enable(2)
        eager_eval(batches[0])
        torch.cuda.synchronize()
        MSDA.profile_enable(False)
        ktimes.update({k: v for k, v in kernel_times(MSDA.profile_collect()).items() if k[0] == "proj"})
        res["eval"] = {"elapsed": elapsed, "per_rank": per_rank, "ktimes": ktimes, "clock_mhz": eval_clock,
                       "graphs": None if graphed_eval is None else {
                           "cached": len(graphed_eval.graphs), "captures_total": graphed_eval.captures,
                           "captures_in_timed_region": graphed_eval.captures - caps0,
                           "replays": graphed_eval.replays, "decode_segments_replayed": graphed_eval.segments_replayed,
                           "decode_chunk": graphed_eval.decode_chunk,
                           "padded_targets_slots": graphed_eval.capacity.slots}}
        if graphed_eval is not None:
            graphed_eval.graphs.clear()
        del step, graphed_eval
        torch.cuda.empty_cache()
        # A/B of the same step with the token loop's three products on the fp32 library GEMMs (GVL_GEMM=f32) instead of
        # gvl_gemm_f16x3 (fp32 operands split into fp16 (hi, residual) pairs that keep 22 significant bits, fp16 MFMA, fp32 accumulation): reported beside
        # `value`, never as `value`
        from gvl_amd.linear import split_gemm_enabled
        if (split_gemm_enabled() and a.dtype == "f32" and not a.no_graph and not a.no_captioner and not a.no_probes):
            os.environ["GVL_GEMM"] = "f32"
            try:
                g2 = GraphedEvalForward(model, criterion, autocast_dtype=ac, decode_chunk=a.decode_chunk)
                for dt in batches:
                    g2(dt)
                el2, _ = timed_loop(g2, batches, a.steps, a.warmup, world, dev)
                res["eval"]["fp32_library_gemms_elapsed"] = el2
                g2.graphs.clear()
                del g2
            finally:
                del os.environ["GVL_GEMM"]
            torch.cuda.empty_cache()
        # A/B of the same step with the transformer layers / base encoder / heads in their PyTorch formulation (library
        # fp32 GEMMs, ATen elementwise) instead of the hand-written inference layers of gvl_amd/layers.py
        from gvl_amd import layers as _layers
        if (_layers.enabled() and a.dtype == "f32" and not a.no_graph and not a.no_probes):
            os.environ["GVL_LAYERS"] = "torch"
            try:
                g3 = GraphedEvalForward(model, criterion, autocast_dtype=ac, decode_chunk=a.decode_chunk)
                for dt in batches:
                    g3(dt)
                el3, _ = timed_loop(g3, batches, a.steps, a.warmup, world, dev)
                res["eval"]["pytorch_layers_elapsed"] = el3
                g3.graphs.clear()
                del g3
            finally:
                del os.environ["GVL_LAYERS"]
            torch.cuda.empty_cache()

        # ... and with BOTH switches off: every Linear on the fp32 library GEMMs (exact fp32 products, no split-fp16 operands
        # anywhere) -- the number for a reader who does not accept 22-bit operands as fp32
        if (_layers.enabled() and split_gemm_enabled() and a.dtype == "f32" and not a.no_graph and not a.no_captioner
                and not a.no_probes):
            os.environ["GVL_LAYERS"] = "torch"
            os.environ["GVL_GEMM"] = "f32"
            try:
                g4 = GraphedEvalForward(model, criterion, autocast_dtype=ac, decode_chunk=a.decode_chunk)
                for dt in batches:
                    g4(dt)
                el4, _ = timed_loop(g4, batches, a.steps, a.warmup, world, dev)
                res["eval"]["strict_fp32_elapsed"] = el4
                g4.graphs.clear()
                del g4
            finally:
                del os.environ["GVL_LAYERS"]
                del os.environ["GVL_GEMM"]
            torch.cuda.empty_cache()

        # ... and the same step under torch.autocast(bfloat16) with the default autocast inference policy ("f16": the same
        # fp32-storage path with ONE fp16 product per fp32 product -- 11-bit operands where bf16 keeps 8; gvl_amd/pdvc.py):
        # a reduced-precision number, reported beside `value`, never as `value`
        if (_layers.enabled() and split_gemm_enabled() and a.dtype == "f32" and not a.no_graph and not a.no_captioner
                and not a.no_probes and _aip() == "f16"):
            g5 = GraphedEvalForward(model, criterion, autocast_dtype=torch.bfloat16, decode_chunk=a.decode_chunk)
            for dt in batches:
                g5(dt)
            el5, _ = timed_loop(g5, batches, a.steps, a.warmup, world, dev)
            res["eval"]["autocast_f16_elapsed"] = el5
            g5.graphs.clear()
            del g5
            torch.cuda.empty_cache()

    # ---------------------------------------------------------------------------------------------- train half
    def train_half():
        model.train()
        # one process: the whole step is ONE hipGraph; several processes: forward/backward graph, eager bucketed RCCL
        # all-reduce of the flat gradient buffer, clip/Adam graph (no collective inside a graph) -- GraphedTrainStep
        if not a.no_graph:
            trainer = GraphedTrainStep(model, criterion, opt, world_size=world,
                                       split_exchange=True if a.split_exchange else None, autocast_dtype=ac,
                                       cap_len_policy=a.cap_len_policy)
        else:
            trainer = TrainStep(model, criterion, opt, world_size=world, autocast_dtype=ac)
        for i_, dt in enumerate(batches):
            o_ = trainer(dt)
            if os.environ.get("GVL_BENCH_TRACE_LOSS"):
                print(f"[trace] set-up call {i_}: loss {float(o_[0]):.4f}", file=sys.stderr, flush=True)
        caps0 = getattr(trainer, "captures", 0)
        MSDA.profile_enable(a.no_graph)
        elapsed, per_rank = timed_loop(trainer, batches, a.steps, a.warmup, world, dev)
        train_clock = [getattr(timed_loop, "last_clock_mhz", None)]
        last = getattr(timed_loop, "last_out", None)
        try:
            last_loss = float(last[0]) if last is not None else None
        except Exception:                      # noqa: BLE001
            last_loss = None
        MSDA.profile_enable(False)
        if a.no_graph:
            ktimes = kernel_times(MSDA.profile_collect())
        else:
            ktimes = instrumented(lambda dt: TrainStep.__call__(trainer, dt))
            train_clock.append(round(MSDA.clock_probe_mhz(dev, 4000), 0))      # ... and behind the instrumented eager steps
        xch = trainer.exchange_times_ms() if hasattr(trainer, "exchange_times_ms") else None
        res["train"] = {"elapsed": elapsed, "per_rank": per_rank, "ktimes": ktimes, "exchange_ms": xch, "clock_mhz": train_clock, "last_loss": last_loss,
                        "graphs": None if a.no_graph else {
                            "cached": len(trainer.graphs), "captures_total": trainer.captures,
                            "captures_in_timed_region": trainer.captures - caps0, "replays": trainer.replays,
                            "padded_targets_slots": trainer.capacity.slots,
                            "padded_caption_width": trainer.capacity.cap_len,
                            "caption_width_policy": trainer.capacity.cap_len_policy,
                            "device_memory_reserved_GB": round(torch.cuda.memory_reserved() / 2 ** 30, 2),
                            "form": ("three graphs (forward + backward to the encoder output | encoder backward | clip + Adam) with "
                                     "the bucketed gradient exchange posted eagerly between them") if trainer.split
                            else ("one graph per caption-width bucket" if trainer.capacity.cap_len_policy == "bucket" else "one graph")}}

    if a.mode in ("both", "train"):
        if world > 1 and "eval" in res:
            # several processes: a failure of the gradient exchange must not take the measured eval line with it -- it is reported
            # in the line instead (one process: raise, loudly)
            try:
                train_half()
            except Exception as e:                 # noqa: BLE001
                res["train_error"] = f"{type(e).__name__}: {e}"[:400]
                print(f"[bench] rank {rank}: train half failed: {res['train_error']}", file=sys.stderr, flush=True)
        else:
            train_half()

    # ---------------------------------------------------------------------------------------------- the line
    src_note = ("per-dispatch stamps (hipExtLaunchKernel events) of the launches inside the timed region" if a.no_graph
                else "median over twelve instrumented eager steps right after the timed region (timed steps are hipGraph replays); "
                     "kernel_us_mean = the plain mean of the same launches")

    def fwd_roofline(ktimes):
        fwd = {k: v for k, v in ktimes.items() if k[0] in ("fwd_t1d_d64", "fwd_generic")}
        dec_key = next((k for k in fwd if k[1] == a.queries and k[2] == B), None)
        if dec_key is None:
            return None
        us, n, us_mean = fwd[dec_key]
        nbytes = msda_bytes(B, S, a.queries, value_bytes=vb_eval)
        achieved = nbytes / (us * 1e-6) / 1e9              # GB/s
        # (the eval forward's launches are the row-maxima variant the inference layers use: "100_f32_amax")
        traffic, traffic_src = traffic_of("100_f32_amax" if "eval" in res else "100_f32", "fwd_dec")
        roof = {"bound": "hbm", "kernel": f"k_{dec_key[0]} (decoder cross-attention launch, Lq={a.queries})",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s",
                "frac": round(achieved / (HBM_PEAK_TBS * 1e3), 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel_us": round(us, 2), "kernel_us_mean": round(us_mean, 2), "launches_timed": n,
                "algorithmic_bytes": nbytes, "source": src_note}
        enc_key = next((k for k in fwd if k[1] == S and k[2] == B), None)
        if enc_key is not None:
            eus, en = fwd[enc_key][:2]
            eb = msda_bytes(B, S, S, value_bytes=vb_eval)
            roof["encoder_launch"] = {"kernel_us": round(eus, 2), "launches_timed": en, "algorithmic_bytes": eb,
                                      "frac": round(eb / (eus * 1e-6) / 1e9 / (HBM_PEAK_TBS * 1e3), 4)}
        return roof

    def bwd_roofline(ktimes):
        bwd = {k: v for k, v in ktimes.items() if k[0] in ("bwd_t1d_d64", "bwd_generic")}
        part = {k: v for k, v in ktimes.items() if k[0] == "sum_partials"}
        out = {}
        for name, Lq in (("decoder", a.queries), ("encoder", S)):
            key = next((k for k in bwd if k[1] == Lq and k[2] == B), None)
            if key is None:
                continue
            us, n, us_mean = bwd[key]
            pk = part.get(("sum_partials", Lq, B))
            extra = pk[0] * pk[1] / n if pk else 0.0
            nbytes = msda_bwd_bytes(B, S, Lq, value_bytes=vb)
            tot = us + extra
            out[name] = {"kernel_us": round(us, 2), "kernel_us_mean": round(us_mean, 2),
                         "partial_sum_us_per_launch": round(extra, 2), "launches_timed": n,
                         "algorithmic_bytes": nbytes, "achieved": round(nbytes / (tot * 1e-6) / 1e9, 1),
                         "frac": round(nbytes / (tot * 1e-6) / 1e9 / (HBM_PEAK_TBS * 1e3), 4)}
        if not out:
            return None
        d = out.get("decoder") or out["encoder"]
        traffic, traffic_src = traffic_of("100_f32", "bwd_dec")
        return {"bound": "hbm", "kernel": "k_bwd_t1d_d64 (+ k_sum_partials where it still runs), decoder launch",
                "achieved": d["achieved"], "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s", "frac": d["frac"],
                "traffic": traffic, "traffic_source": traffic_src, "launches": out, "source": src_note}

    workload = (f"cfgs/{a.cfg}.yml PDVC B={B}/GPU T={a.T} L=4 Q={a.queries}, "
                + ("captioner off (diagnostic)" if a.no_captioner else
                   f"LSTM-DSA greedy captioning of up to {opt.max_caption_len} tokens (the reference's extra LSTM step "
                   f"after the last token is dead code -- never read -- and is not evaluated)")
                + ", set criterion + Hungarian matcher; "
                + ("one fixed batch, 3 events per video" if a.fixed_layout else
                   f"{len(batches)} rotating batches with 0-10 events per video and 3-20-word captions"))
    from gvl_amd.linear import split_gemm_enabled as _sge
    from gvl_amd.pdvc import autocast_inference_policy, autocast_training_policy
    _pol, _tpol = autocast_inference_policy(), autocast_training_policy()
    island = a.dtype == "bf16" and _pol != "bf16"                # eval forward under autocast = the fp32-storage path
    gemm16_on = _sge() and (a.dtype == "f32" or island)
    from gvl_amd import layers as _lay
    _lay_on = _lay.enabled() and (a.dtype == "f32" or island)
    line = {"n_gpus": world, "steps": a.steps, "warmup": a.warmup, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.dtype == "f32" else (
                "torch.autocast(bfloat16): "
                + {"f16": "train step on the hand-written fp32-storage training path with ONE fp16 product per fp32 product in the "
                          "forward, input-gradient and weight-gradient products (11-bit operands, fp32 accumulate, fp32 master "
                          "weights; GVL_AUTOCAST_TRAINING=bf16 for torch's bf16 formulation); ",
                   "fp32": "train step on the hand-written fp32-storage training path, exact products (GVL_AUTOCAST_TRAINING=fp32); ",
                   "bf16": "train step on bf16 storage + bf16 library GEMMs (f32 accumulate / locations / captioner); "}[_tpol]
                + (("eval forward on the hand-written fp32-storage inference path with ONE fp16 product per fp32 product "
                    "(11-bit operands, fp32 accumulate; GVL_AUTOCAST_INFERENCE=fp32 for the exact products, =bf16 for bf16 "
                    "storage)" if _pol == "f16" else
                    "eval forward as an fp32 island on the hand-written inference path (GVL_AUTOCAST_INFERENCE=fp32)")
                   if island else "eval forward on bf16 storage + bf16 GEMMs")),
            "data": "synthetic",
            "config": {"workload": workload,
                       "library_gemm_selection": "gvl_amd/tunableop_mi355x.csv" if tuned else "hipBLASLt default",
                       "token_loop_gemms": ("gvl_gemm_f16x3: fp32 operands split into fp16 (hi, 2^11 residual) pairs that keep 22 of fp32's 24 significant bits (operand_bits = 22), "
                                            "3 fp16-MFMA partial products, fp32 accumulation -- error vs fp64 "
                                            "below the fp32 GEMM's (tests/test_gpu_gemm16.py); per token: h2att(h) in the greedy "
                                            "reduction's launch, both halves of the LSTM gate product + cell in one launch "
                                            "(k_gates_f16x3), vocabulary product with argmax / log-sum-exp fused (k_vocab_f16x3)")
                       if gemm16_on else "hipBLASLt fp32",
                       "inference_layers": ("gvl_amd/layers.py: every Linear of the encoder / decoder layers, the box MLP, the "
                                            "captioner's per-forward constants and the base encoder's conv1d levels on "
                                            "gvl_linear_f16x3_f32 (fp32 A split in the load path, fused bias / ReLU / residual / "
                                            "mask / row-maxima epilogues), LayerNorm / GroupNorm / attention core / refinement "
                                            "as hand-written kernels") if _lay_on else "PyTorch (GVL_LAYERS=torch)",
                       "global_batch": world * B,
                       "parallelism": f"dp{world} (videos sharded; eval: no data-path collective, train: RCCL gradient "
                                      f"all-reduce)"},
            "rccl_ranks": world if (world > 1 and backend == "nccl") else (0 if world == 1 else f"{world} ({backend})")}
    kernels_us = {}

    def ktag(k):                         # (kernel, meta_a, meta_b): the GEMM launches differ in meta_b (N) only
        return f"{k[0]}[{k[1]}x{k[2]}]" if k[0] == "gemm_f16x3" else f"{k[0]}[{k[1]}]"
    if "eval" in res:
        e = res["eval"]
        ms = e["elapsed"] * 1e3 / a.steps
        line.update({"metric": "videos/sec (eval fwd)", "value": round(world * B * a.steps / e["elapsed"], 3),
                     "unit": "videos/s", "ms_per_step": round(ms, 3), "higher_is_better": True,
                     "roofline": fwd_roofline(e["ktimes"]), "eval_graphs": e["graphs"],
                     "eval_seconds_per_rank": [round(x, 4) for x in e["per_rank"]]})
        kernels_us["eval"] = {ktag(k): {"us": round(v[0], 2), "n": v[1]} for k, v in e["ktimes"].items()
                              if not k[0].startswith("fwd_")}
        if "pytorch_layers_elapsed" in e:
            e3 = e["pytorch_layers_elapsed"]
            line["eval_with_pytorch_layers"] = {
                "value": round(world * B * a.steps / e3, 3), "ms_per_step": round(e3 / a.steps * 1e3, 3),
                "note": "same run, GVL_LAYERS=torch: encoder / decoder layers, base encoder and heads through PyTorch "
                        "(library fp32 GEMMs + ATen kernels) instead of gvl_amd/layers.py (gvl_linear_f16x3_f32 with fused "
                        "epilogues, gvl_layer_norm_rows_f32, gvl_mha_core_f32, ...); token loop unchanged"}
        if "strict_fp32_elapsed" in e:
            e4 = e["strict_fp32_elapsed"]
            line["eval_strict_fp32"] = {
                "value": round(world * B * a.steps / e4, 3), "unit": "videos/s", "ms_per_step": round(e4 * 1e3 / a.steps, 3),
                "note": "same run, GVL_LAYERS=torch AND GVL_GEMM=f32: every Linear of the step an exact-fp32 library GEMM "
                        "(hipBLASLt), no split-fp16 operands anywhere; the hand-written sampling / attention / criterion / "
                        "matcher kernels (exact fp32 arithmetic) stay"}
        if "autocast_f16_elapsed" in e:
            e5 = e["autocast_f16_elapsed"]
            line["eval_under_autocast"] = {
                "value": round(world * B * a.steps / e5, 3), "unit": "videos/s", "ms_per_step": round(e5 * 1e3 / a.steps, 3),
                "policy": "f16",
                "note": "REDUCED PRECISION, not comparable with `value`: the same step under torch.autocast(bfloat16) with the "
                        "default policy of gvl_amd.pdvc.autocast_inference_policy -- the fp32-storage path with one fp16 "
                        "matrix-core product per fp32 product instead of three (operands rounded to 11 bits at their row scale, "
                        "bf16 keeps 8; fp32 accumulation, fp32 activations).  Logits within 1.4e-3 of the fp32 forward where "
                        "the bf16 library route sits 1.1e-1 away (tools/f16_policy_probe.py)"}
        if "fp32_library_gemms_elapsed" in e:
            e2 = e["fp32_library_gemms_elapsed"]
            line["eval_with_fp32_library_gemms"] = {
                "value": round(world * B * a.steps / e2, 3), "unit": "videos/s", "ms_per_step": round(e2 * 1e3 / a.steps, 3),
                "note": "the same step with GVL_GEMM=f32: the captioner's three token-loop products on hipBLASLt fp32 "
                        "GEMMs + gvl_greedy_step_f32 over written logits (round 2's path until gvl_gemm_f16x3)"}
    if "train_error" in res:
        line["train_error"] = res["train_error"]
    if "train" in res:
        t_ = res["train"]
        ms = t_["elapsed"] * 1e3 / a.steps
        line.update({"train_step_ms": round(ms, 3), "train_videos_per_s": round(world * B * a.steps / t_["elapsed"], 3),
                     "train_last_timed_step_loss": t_.get("last_loss"),
                     "train_roofline": bwd_roofline(t_["ktimes"]), "train_graphs": t_["graphs"],
                     "train_seconds_per_rank": [round(x, 4) for x in t_["per_rank"]]})
        if t_.get("exchange_ms"):
            # the eager gradient exchange between the captured graphs (rank 0's view): total = first bucket posted -> all
            # buckets reduced, exposed = the part that is not hidden under the encoder's backward
            line["grad_exchange_ms_per_step"] = {"total": round(t_["exchange_ms"][0], 3),
                                                 "exposed": round(t_["exchange_ms"][1], 3),
                                                 "bytes": int(sum(p_.numel() for p_ in model.parameters() if p_.requires_grad) * 4)}
        kernels_us["train"] = {ktag(k): {"us": round(v[0], 2), "n": v[1]} for k, v in t_["ktimes"].items()}
        if "eval" not in res:
            line.update({"metric": "train-step ms", "value": round(ms, 3), "unit": "ms", "ms_per_step": round(ms, 3),
                         "higher_is_better": False, "roofline": line["train_roofline"]})
    line["kernels_us"] = kernels_us
    line["shader_clock_mhz"] = {"after_eval_timed_region": (res.get("eval") or {}).get("clock_mhz"),
                                "after_train_timed_region_and_instrumented_steps": (res.get("train") or {}).get("clock_mhz"),
                                "note": "gvl_clock_probe: cycle counter against the 100 MHz wall clock over a chain of dependent "
                                        "FMAs queued right behind the region (MI355X max 2400)"}
    if line.get("roofline") is not None and rank == 0 and a.T != 512 and a.dtype == "f32" and "eval" in res and not a.no_probes:
        line["roofline"]["cfg_L_launch"] = kernel_probe(dev, B)
        line["roofline"]["B64_launch"] = kernel_probe(dev, 64, T=a.T, Q=a.queries)
        line["roofline"]["B256_launch"] = kernel_probe(dev, 256, T=a.T, Q=a.queries)      # latency-free sweep 16 / 64 / 256
    if line.get("train_roofline") is not None and rank == 0 and a.T != 512 and a.dtype == "f32" and not a.no_probes:
        line["train_roofline"]["cfg_L_launch"] = kernel_probe(dev, B, backward=True)
        line["train_roofline"]["B64_launch"] = kernel_probe(dev, 64, T=a.T, Q=a.queries, backward=True)
        line["train_roofline"]["B256_launch"] = kernel_probe(dev, 256, T=a.T, Q=a.queries, backward=True)
    if rank == 0 and "eval" in res and not a.no_captioner and a.dtype == "f32" and not a.no_probes:
        with torch.no_grad():
            lib = gemm_probe(model, dev, B * a.queries)
        head = model.caption_head[-1]
        V, Kd, rows = head.logit.weight.shape[0], head.logit.weight.shape[1], B * a.queries
        mine = [v for k, v in res["eval"]["ktimes"].items() if k[0] == "gemm_f16x3" and k[2] == V]
        if gemm16_on and mine:
            us = mine[0][0]
            tf16 = 3 * 2.0 * rows * V * Kd / (us * 1e-6) / 1e12
            # which kernel form serves it: the dispatch rule of gvl_gemm_f16x3_argmax_f32 (gvl_gemm16.hip), GVL_VOCAB_FORM overrides
            cus = torch.cuda.get_device_properties(dev).multi_processor_count // 8 * 8 or 8
            cost_v = -(-(-(-V // 256) * -(-rows // 320)) // cus) * 23
            cost_m = -(-(-(-V // 128) * -(-rows // 256)) // cus) * 10
            form = os.environ.get("GVL_VOCAB_FORM", "")[:1]
            big = Kd % 64 == 0 and Kd >= 128 and rows >= 1024 and (form == "v" or (form != "m" and cost_v < cost_m))
            kname = "k_vocab_f16x3 (256x320 tiles, one accumulator)" if big else "k_gemm_f16x3_m16<argmax>"
            line["dominant_gemm"] = {
                "kernel": f"{kname} {rows}x{Kd}x{V} (vocabulary product of every token step, argmax / "
                          f"log-sum-exp fused, logits never written), hand-written",
                "note": "the matrix pipe is power-limited under this load (clock 1.87 GHz, issue throttled by the operands' "
                        "switching activity): with real operands and NO data movement the same MFMA stream reaches 0.63 of "
                        "the nominal peak (DESIGN_LOG.md 4.4, tools/vocab_clocks.sh); profiles/r06_gemm16_pmc.txt: MFMA pipe busy 0.66 of a "
                        "wavefront's resident cycles at >= 1.76 GHz",
                "bound": "mfma", "achieved": round(tf16, 1), "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (fp16 MFMA, 3 "
                "partial products per fp32 product)", "frac": round(tf16 / F16_MFMA_PEAK_TFLOPS, 4),
                "fp32_equivalent_tflops": round(tf16 / 3, 1), "kernel_us": round(us, 1), "launches_timed": mine[0][1],
                "source": "in-library stamps of two instrumented eager steps after the timed region",
                "fp32_library_gemm_same_shape": lib}
        else:
            line["dominant_library_gemm"] = lib
    if rank == 0 and world == 1 and not a.no_probes and not a.no_graph and a.dtype == "f32":
        # the same kernels' durations INSIDE the replayed graphs, from a rocprofv3 kernel trace of a short child run of this script
        # (the eager stamps above read 3-10 % shorter than the launches the timed region actually replays)
        ig = in_graph_kernel_us(a, "both" if ("eval" in res and "train" in res) else ("eval" if "eval" in res else "train"))
        for key, kern, bytes_key in (("roofline", "fwd", None), ("train_roofline", "bwd", None)):
            blk = line.get(key)
            if blk is None or (key == "roofline" and "eval" not in res):
                continue
            got = ig.get(kern) if isinstance(ig, dict) else None
            blk["kernel_us_in_graph"] = got["decoder_us"] if got else None
            nbytes = blk.get("algorithmic_bytes") or ((blk.get("launches") or {}).get("decoder") or {}).get("algorithmic_bytes")
            if got and nbytes:
                blk["frac_in_graph"] = round(nbytes / (got["decoder_us"] * 1e-6) / 1e12 / HBM_PEAK_TBS, 4)
                blk["in_graph_launches"] = got
            else:
                blk["in_graph_note"] = ig.get("error") if isinstance(ig, dict) else str(ig)
    if rank == 0 and world == 1 and not a.no_cpu_baseline and "eval" in res:
        line["cpu_baseline"] = cpu_baseline(model, opt, a.T)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()                    # rank 0 ran the supplementary probes; leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
