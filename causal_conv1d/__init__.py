"""Import-compatible facade for ``causal_conv1d`` (causal-conv1d==1.4.0, /root/reference/requirements.txt:12)."""
__version__ = "1.4.0+omnimamba_amd"
from omnimamba_amd.causal_conv1d import causal_conv1d_fn, causal_conv1d_update  # noqa: F401
