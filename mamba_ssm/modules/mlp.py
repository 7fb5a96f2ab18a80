"""``mamba_ssm.modules.mlp.GatedMLP`` (imported by /root/reference/models/stage2/mixer_seq_simple.py:18; built by create_block when
``d_intermediate > 0`` -- no shipped OmniMamba config does, config_mamba.py:7).  The published module restated: fc1 to twice the hidden
width, y * act(gate) on the two halves, fc2 back; hidden width 8/3 of the input rounded up to ``multiple_of``.  Both projections are
library GEMMs through ``omnimamba_amd.linear`` (token-split weight gradient, formed before the input gradient)."""
import torch.nn as nn
import torch.nn.functional as F

from omnimamba_amd.linear import linear


class GatedMLP(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, activation=F.silu, bias=False, multiple_of=128,
                 device=None, dtype=None):
        super().__init__()
        fk = {"device": device, "dtype": dtype}
        out_features = in_features if out_features is None else out_features
        hidden = int(8 * in_features / 3) if hidden_features is None else hidden_features
        hidden = (hidden + multiple_of - 1) // multiple_of * multiple_of
        self.fc1 = nn.Linear(in_features, 2 * hidden, bias=bias, **fk)
        self.activation = activation
        self.fc2 = nn.Linear(hidden, out_features, bias=bias, **fk)

    def forward(self, x, task_types=None):
        # (task_types: the reference's Block passes it, block.py:138-143 -- the published module would refuse it; it has no LoRA to switch)
        y, gate = linear(x, self.fc1.weight, self.fc1.bias).chunk(2, dim=-1)
        return linear(y * self.activation(gate), self.fc2.weight, self.fc2.bias)
