"""``mamba_ssm.ops.selective_scan_interface``."""
from omnimamba_amd.selective_scan import mamba_inner_fn, selective_scan_fn, selective_scan_ref  # noqa: F401
