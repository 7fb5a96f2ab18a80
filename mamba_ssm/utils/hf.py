"""``load_config_hf`` / ``load_state_dict_hf`` (imported at mixer_seq_simple.py:20, used only by from_pretrained):
local-directory loading only -- there is no hub access on the GPU box."""
import json
import os

import torch


def load_config_hf(model_name):
    path = os.path.join(model_name, "config.json")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: only local checkpoints are supported (no network)")
    return json.load(open(path))


def load_state_dict_hf(model_name, device=None, dtype=None):
    path = os.path.join(model_name, "pytorch_model.bin")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: only local checkpoints are supported (no network)")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if dtype is not None:
        sd = {k: v.to(dtype=dtype) for k, v in sd.items()}
    return {k: v.to(device=device) for k, v in sd.items()}
