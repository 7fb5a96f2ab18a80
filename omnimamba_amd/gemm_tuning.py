"""Plain library GEMMs (in_proj / out_proj and their gradients) stay with hipBLASLt / rocBLAS; this module only picks
WHICH library solution runs, through PyTorch's TunableOp, from a results file recorded on an MI355X
(`tools/tune_gemms.py`, 1.3B block shapes: tokens 32768 x {2048 -> 8512, 4096 -> 2048} and their dgrad / wgrad forms).
Measured effect on real activations (`tools/gemm_ab.py`): 1-3 % of the GEMM time -- TunableOp times its candidates on
idle-clock buffers, so the 0.6 ms it records for the in_proj forward is 0.96 ms under load.  The weight-gradient GEMMs
are helped more by the token split in `omnimamba_amd/linear.py`.

The file carries validators (PyTorch / ROCm / hipBLASLt versions, gfx arch); on any mismatch TunableOp ignores it and
the default solutions run, so enabling this is always safe.
"""
from __future__ import annotations

import os

import torch

_DEFAULT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gemm_gfx950_block_1p3b.csv")


def use_tuned_gemms(path: str | None = None, tune_missing: bool = False) -> bool:
    """Enable TunableOp with the recorded solutions.  `tune_missing=True` additionally tunes GEMM shapes that are not in
    the file the first time they run (seconds per shape) -- off by default so a timed run never pays for tuning."""
    path = path or os.environ.get("OMK_GEMM_TUNING_FILE", _DEFAULT)
    if os.environ.get("OMK_GEMM_TUNING", "1") == "0" or not torch.cuda.is_available() or not os.path.exists(path):
        return False
    tun = torch.cuda.tunable
    tun.enable(True)
    tun.tuning_enable(bool(tune_missing))
    if hasattr(tun, "record_untuned_enable"):
        tun.record_untuned_enable(False)
    try:
        torch._C._cuda_tunableop_write_file_on_exit(False)   # never write next to the caller's cwd
    except Exception:
        pass
    ok = bool(tun.read_file(path))
    if not ok and not tune_missing:
        tun.enable(False)
    return ok
