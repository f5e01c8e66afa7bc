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
__version__ = "0.1.0"
