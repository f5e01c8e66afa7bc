"""gvl_amd -- MI355X-native (gfx950) implementation of GVL's deformable-transformer hot path.

Layout mirrors the reference's ``pdvc`` package for the files on the path:

    gvl_amd.MultiScaleDeformableAttention   <- the native extension module of pdvc/ops (vision.cpp:13-16)
    gvl_amd.ops.functions                   <- pdvc/ops/functions/ms_deform_attn_func.py
    gvl_amd.ops.modules                     <- pdvc/ops/modules/{ms_deform_attn,ms_deform_attn_for_caption}.py
    gvl_amd.deformable_transformer          <- pdvc/deformable_transformer.py
    gvl_amd.matcher                         <- pdvc/matcher.py
    gvl_amd.pdvc                            <- pdvc/pdvc.py (build / PDVC.forward contract)

All compute on the path goes through libgvl_msda.so (hand-written HIP, C ABI in include/gvl_msda.h).  There is no
CPU fallback: importing works anywhere, calling an op without the built library or a ROCm device raises.
"""
import os as _os

# ROCm 7.2 hipGraph replays: with the runtime's AQL "packet capture" fast path (default on) a captured step that contains
# non-kernel nodes (PyTorch's hipMemsetAsync of reduction semaphores, device-to-device copies) can replay them out of order
# with the kernel packets once the queue has gone idle -- seen as garbage bias gradients (column sums) and NaN parameters in
# the captured bf16 train step two replays after ANY host synchronisation (tools/bf16_nan_repro3.py; gone with the switch
# below, eager steps never affected; cost: +0.6 % on the fp32 train step, none on the eval forward).  The runtime reads the
# variable when it initialises, so it has to be in the environment before the first HIP call of the process: import gvl_amd
# (or bench.py) before touching the GPU, or export it yourself.  An explicit setting in the environment wins.
import sys as _sys

_prev = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")
_torch = _sys.modules.get("torch")
_hip_up = bool(_torch is not None and hasattr(_torch, "cuda") and _torch.cuda.is_initialized())
# the setting only works if it is in place before the process's first HIP call.  Remember when it was NOT (the
# runtime already initialised without it, or an explicit non-zero setting): the captured steps of gvl_amd.parallel refuse to
# capture then (graph_replay_guard) instead of silently corrupting a training run.
GRAPH_REPLAY_UNSAFE = _prev != "0" and (_hip_up or _prev is not None)
if _prev is None:
    _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"


def graph_replay_guard(what):
    """raise if hipGraph replays of a step with non-kernel nodes are known to be unsafe in this process (see above);
    GVL_ALLOW_GRAPH_PACKET_CAPTURE=1 overrides (a runtime without the bug)"""
    if GRAPH_REPLAY_UNSAFE and _os.environ.get("GVL_ALLOW_GRAPH_PACKET_CAPTURE", "") != "1":
        raise RuntimeError(
            f"{what}: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was not in the environment when the HIP runtime initialised (it was "
            f"{_prev!r}).  On ROCm 7.2 a replayed hipGraph can then execute its memset / copy nodes out of order (NaN parameters "
            "two replays after a host synchronisation).  Export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before launching the process, or "
            "import gvl_amd before the first CUDA call; GVL_ALLOW_GRAPH_PACKET_CAPTURE=1 overrides this check.")

__version__ = "0.1.0"
