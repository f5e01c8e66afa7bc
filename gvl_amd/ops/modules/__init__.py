from .ms_deform_attn import MSDeformAttn  # noqa: F401
from .ms_deform_attn_for_caption import MSDeformAttnCap  # noqa: F401
