from omnimamba_amd.mamba2 import Mamba2  # noqa: F401
