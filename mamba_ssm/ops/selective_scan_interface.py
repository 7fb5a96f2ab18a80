from omnimamba_amd.selective_scan import selective_scan_fn  # noqa: F401
