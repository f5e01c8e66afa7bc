"""Library-GEMM selection for the dense layers of the path (value / output / FFN / MHA / LSTM / vocabulary GEMMs).

These are plain library GEMMs (hipBLASLt / rocBLAS through PyTorch); what is chosen here is WHICH library kernel runs
for each shape.  PyTorch's TunableOp benchmarks every hipBLASLt / rocBLAS solution per GEMM signature; doing that at
run time costs ~1 s per new shape, so the results for the BASELINE shapes are tuned once on an MI355X
(``python tools/tune_gemms.py``) and shipped as ``gvl_amd/tunableop_mi355x.csv``.  ``enable_tuned_gemms()`` loads the
file with tuning disabled: shapes in the file get their measured-fastest kernel (the three per-token captioner GEMMs
gain 5-25 %), every other shape keeps the library default.  The CSV carries validators (ROCm / hipBLASLt version,
gfx arch); PyTorch ignores it if they do not match the running stack.
"""
import os

import torch

DEFAULT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_mi355x.csv")


def enable_tuned_gemms(path=DEFAULT_FILE, tune_missing=False):
    """Returns True when the tuned table was loaded."""
    if not torch.cuda.is_available() or not os.path.exists(path):
        return False
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(bool(tune_missing))
    try:
        ok = bool(torch.cuda.tunable.read_file(path))
        # TunableOp rewrites "its" file at process exit; point that at a scratch path, never at the shipped table
        torch.cuda.tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "gvl_tunableop_scratch.csv"), True)
        return ok
    except Exception:
        torch.cuda.tunable.enable(False)
        return False
