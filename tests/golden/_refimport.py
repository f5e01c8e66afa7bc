"""Import shim for the upstream reference at /root/reference (build container only).

Used ONLY by tests/golden/make_golden*.py to emit the committed golden vectors.
Nothing here is imported by the product, the tests or the bench at run time, and
nothing from /root/reference is copied: the reference is imported in place,
its modules that are never *used* on the hot path (torchvision, colorlog,
pycocoevalcap, transformers.AdamW) are replaced by empty stubs.
"""
import importlib.machinery
import os
import sys
import types

REF = os.environ.get("GVL_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install(full=False):
    """Make `import pdvc...` resolve to the reference. full=True also allows pdvc.pdvc."""
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference not found at {REF}; golden vectors can only be "
                           "regenerated in the build container")
    import torch
    if full:
        # transformers must finish its own lazy imports BEFORE torchvision is stubbed,
        # and it re-registers sys.modules['transformers'] while doing so.
        import transformers  # noqa: F401
        from transformers import (GPT2Tokenizer, GPT2LMHeadModel,  # noqa: F401
                                  get_linear_schedule_with_warmup, AutoModel, BertConfig)
        from transformers.models.bert.modeling_bert import BertEncoder  # noqa: F401
        sys.modules["transformers"].AdamW = torch.optim.AdamW  # removed in transformers 5
    if REF not in sys.path:
        sys.path.insert(0, REF)
    _stub("torchvision", __version__="0.25.0")
    _stub("torchvision.ops")
    _stub("torchvision.ops.boxes", box_area=None)
    _stub("colorlog")
    for n in ("pycocoevalcap", "pycocoevalcap.meteor", "pycocoevalcap.bleu"):
        _stub(n)
    _stub("pycocoevalcap.meteor.meteor", Meteor=object)
    _stub("pycocoevalcap.bleu.bleu", Bleu=object)
    if not full:
        # pdvc/CaptioningHead/__init__.py pulls in HF GPT-2; expose the package
        # directory without executing its __init__ so LSTM_DSA imports alone.
        pkg = _stub("pdvc.CaptioningHead")
        pkg.__path__ = [os.path.join(REF, "pdvc", "CaptioningHead")]
