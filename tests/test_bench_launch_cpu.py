"""`python bench.py --gpus N` (N > 1) without a launcher starts its ranks itself (bench.self_launch): here, without a GPU,
only the control flow can be checked -- the parent starts a child job, never initialises the GPU itself, and a failing
rank (no device on this box) comes back as a non-zero exit code instead of a hang or a silent 0.  The passing case runs
on the GPU box (tests/test_gpu_bench_dp.py, launcher=False)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="failure path of a box without GPUs")
def test_self_launch_propagates_a_failing_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--launch-timeout", "240"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert out.returncode not in (0, 124), (out.returncode, out.stderr[-1500:])
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # the children were started through the launcher with the driver's arguments
    assert "torch.distributed" in out.stderr or "torchrun" in out.stderr or "ChildFailedError" in out.stderr, out.stderr[-1500:]


def test_single_gpu_invocation_does_not_self_launch():
    import bench
    import inspect
    src = inspect.getsource(bench.main)
    assert '"WORLD_SIZE" not in os.environ and a.gpus > 1' in src
    # the branch sits before the first device call of main()
    assert src.index("self_launch(") < src.index("torch.cuda.set_device")
