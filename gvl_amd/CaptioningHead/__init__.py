"""Captioning heads on the path.  Only the LSTM-DSA head (``caption_decoder_type: standard``, used by every
BASELINE config) is built; the reference's other heads (light LSTM, transformer, GPT-2, puppet) are out of scope
 and raise here instead of silently degrading."""
from .LSTM_DSA import LSTMDSACaptioner


def build_captioner(opt):
    if opt.caption_decoder_type == 'standard':
        return LSTMDSACaptioner(opt)
    raise ValueError(f"gvl_amd builds caption_decoder_type='standard' only (got {opt.caption_decoder_type!r}); "
                     "the other reference heads are outside the accelerated path")
