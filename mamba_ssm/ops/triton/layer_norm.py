from omnimamba_amd.layer_norm import RMSNorm, layer_norm_fn, rms_norm_fn, LayerNormFn  # noqa: F401
