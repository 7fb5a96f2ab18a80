"""``mamba_ssm.modules.mamba_simple`` (imported at /root/reference/models/stage2/mixer_seq_simple.py:16): the Mamba-1 mixer."""
from omnimamba_amd.mamba_simple import Mamba  # noqa: F401
