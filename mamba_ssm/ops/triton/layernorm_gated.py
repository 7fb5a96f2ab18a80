from omnimamba_amd.layernorm_gated import RMSNorm, rmsnorm_fn  # noqa: F401
