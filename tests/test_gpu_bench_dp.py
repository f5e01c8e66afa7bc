"""bench.py's N > 1 control flow (rendezvous, barriers, max-over-ranks timing, one JSON line from rank 0) and the
data-parallel train step (forward/backward graph, bucketed all-reduce, clip/Adam graph) with REAL collectives: two ranks
share the one GPU of the test box and talk over gloo (RCCL needs one GPU per rank; the driver's multi-GPU run covers it)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", ["both", "train"])
def test_two_ranks_on_one_gpu(mode):
    env = dict(os.environ, GVL_DIST_BACKEND="gloo", GVL_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--batch", "4", "--queries", "40", "--rotate", "3", "--no-cpu-baseline", "--no-probes"] + (["--mode", "train"] if mode == "train" else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["global_batch"] == 8
    assert d["value"] > 0 and d["scaling"] == "weak" and d["train_step_ms"] > 0
    assert len(d["train_seconds_per_rank"]) == 2 and d["train_graphs"]["captures_in_timed_region"] == 0
    if mode == "both":
        assert d["metric"].startswith("videos/sec") and d["eval_graphs"]["cached"] == 1
