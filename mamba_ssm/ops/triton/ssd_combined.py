from omnimamba_amd.ssd_combined import mamba_chunk_scan_combined, mamba_split_conv1d_scan_combined  # noqa: F401
