"""``mamba_ssm.modules.mamba_simple.Mamba`` is imported by the reference (mixer_seq_simple.py:16) but only constructed
when ssm_cfg.layer == "Mamba1", which no shipped OmniMamba config selects (models/stage2/config_mamba.py:16)."""
import torch.nn as nn


class Mamba(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("Mamba-1 mixer module is outside the OmniMamba hot path (SURVEY.md section 8 a13: only its "
                                  "selective_scan_fn op is provided, see omnimamba_amd.selective_scan)")
