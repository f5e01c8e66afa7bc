from .ms_deform_attn_func import MSDeformAttnFunction, MSDeformAttnPadFunction  # noqa: F401
