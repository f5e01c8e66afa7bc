from .ms_deform_attn_func import (MSDeformAttnFunction, MSDeformAttnPadFunction, MSDASampleFunction,  # noqa: F401
                                   MSDeformAttnFusedFunction)
