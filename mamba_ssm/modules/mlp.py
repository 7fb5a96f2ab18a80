"""Import-compat stub: d_intermediate == 0 in every OmniMamba config, so GatedMLP is never built (block.py:46-52)."""
import torch.nn as nn


class GatedMLP(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("GatedMLP is outside the OmniMamba hot path (d_intermediate == 0)")
