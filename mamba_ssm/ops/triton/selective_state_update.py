from omnimamba_amd.selective_state_update import selective_state_update  # noqa: F401
