"""Import-compat stub: attention layers are never built by OmniMamba (attn_layer_idx is empty, config_mamba.py:17)."""
import torch.nn as nn


class MHA(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("MHA is outside the OmniMamba hot path (attn_layer_idx is always empty)")
