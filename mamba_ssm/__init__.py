"""Import-compatible facade: answers to the ``mamba_ssm`` module paths OmniMamba imports
(/root/reference/models/stage2/mixer_seq_simple.py:15-20,30 ; models/stage2/block.py:10) and re-exports the
MI355X implementations in ``omnimamba_amd``.  (The sub-package is called ``ops.triton`` only because that is the
path the reference imports; nothing here is Triton.)"""
__version__ = "2.2.2+omnimamba_amd"
from omnimamba_amd.selective_scan import selective_scan_fn  # noqa: F401
from omnimamba_amd.mamba2 import Mamba2  # noqa: F401
