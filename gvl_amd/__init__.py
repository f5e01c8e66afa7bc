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
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

__version__ = "0.1.0"
