"""The data-parallel captured train step at the real cfg A dimensions with world size 2: two fresh processes
(tests/dp_graphed_worker.py) run three GraphedTrainStep updates on their shards of three global batches with uneven events
(videos without events on both shards) -- collectives over gloo on the one shared GPU of the test box, RCCL with two GPUs.
  * both ranks hold IDENTICAL parameters after every step (bit for bit: the same averaged gradients, the same update);
  * they equal, to fp32 summation order, what ONE process computes when it walks the shards in turn, normalises the criterion by
    the mean target count over the shards (criterion.py:178-180), averages the gradients and applies one clip + Adam.
(The single-process step on the CONCATENATED batch is not that oracle: the caption and counter losses are means over a batch's
own tokens / videos, so a mean of per-shard means differs from the mean over the union whenever the shards' counts differ -- a
property of the reference's losses under any data parallelism, not of this implementation.)"""
import os
import signal
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_hold_identical_parameters_and_match_the_serial_data_parallel_step(tmp_path):
    import torch
    two_gpus = torch.cuda.device_count() >= 2
    port = str(_free_port())
    worker = os.path.join(ROOT, "tests", "dp_graphed_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    procs = []
    for r in range(2):
        e = dict(env, GVL_DIST_BACKEND="nccl" if two_gpus else "gloo", GVL_TEST_DEVICE=str(r if two_gpus else 0))
        procs.append(subprocess.Popen([sys.executable, worker, "dp", str(r), "2", port, str(tmp_path / f"rank{r}.npz")], env=e, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True))
    serial = subprocess.Popen([sys.executable, worker, "serial", "0", "2", port, str(tmp_path / "serial.npz")], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    outs = []
    try:
        for p in procs + [serial]:
            outs.append(p.communicate(timeout=420)[0])
    except subprocess.TimeoutExpired:
        for p in procs + [serial]:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        raise
    for p, o in zip(procs + [serial], outs):
        assert p.returncode == 0, o[-4000:]
    r0, r1, ser = (dict(np.load(tmp_path / n)) for n in ("rank0.npz", "rank1.npz", "serial.npz"))
    assert int(r0["captures"][0]) >= 1
    keys = [k for k in r0 if k.startswith("s")]
    assert len(keys) > 20
    for k in keys:
        assert np.array_equal(r0[k], r1[k]), k                                   # the replicas stay identical, bit for bit
    init_moved = 0
    for k in keys:
        if k.endswith("checksum"):
            continue
        a, b = r0[k].astype(np.float64), ser[k].astype(np.float64)
        step = int(k[1])
        # Adam's normalised update: every element moves by about lr per step whatever its gradient's size, so elements whose
        # gradient is fp32 noise can differ by a whole step; compare in units of the update size (lr = 1e-4)
        assert np.abs(a - b).max() <= 2.5e-4 * (step + 1), (k, float(np.abs(a - b).max()))
        frac_close = float((np.abs(a - b) <= 2e-6).mean())
        # The first step starts from identical parameters: the replicas and the serial step then differ by fp32 noise alone (the
        # captioner's float atomics) and almost every element agrees.  Later steps start from parameters that differ by that noise,
        # which can tip a DISCRETE decision of the step -- a Hungarian near-tie between the (nearly identical, randomly initialised)
        # queries, a sample crossing a frame boundary: then 5-17 % of a weight's elements move by a different fraction of an Adam
        # step, reproducibly the same elements (profiles/r06_bwd_experiments.txt: bitwise-deterministic kernels, bit-identical
        # ranks, levels 0.83 / 0.93 / 0.98 / 1.00 depending on data seed and on numerically equivalent builds).  What a stale
        # operand in a replayed graph would do is covered where it cannot be confused with that:
        # tests/test_gpu_layout_independent.py::test_replayed_step_rebuilds_every_operand_plane.
        assert frac_close >= (0.97 if step == 0 else 0.75), (k, frac_close)
        init_moved += 1
    assert init_moved > 20
