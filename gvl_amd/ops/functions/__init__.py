from .ms_deform_attn_func import (MSDeformAttnFunction, MSDeformAttnPadFunction, MSDASampleFunction,  # noqa: F401
                                   MSDeformAttnFusedFunction, ms_deform_attn_core_pytorch)
